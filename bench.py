#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

One "step" = one pass of the hot path over one batch of synthetic blobs that
are already resident in HBM: `blob_to_kzg_commitment` on 4096 blobs per GPU
(BASELINE.json configs[1]).  With --gpus N the driver launches N ranks (one per
GPU); every rank commits its own 4096 blobs (weak scaling, blobs are
independent) and the 48-byte commitments are all-gathered with RCCL.

Prints ONE JSON line on rank 0 (contract in the task description) carrying
`roofline` (dominant kernel k_msm_fixed28, HIP-event timed inside the library on
the stream it runs on) and `cpu_baseline` (the C port of the reference's CPU
algorithm, oracle/cport, timed on the host cores on a bounded sample).
Secondary workloads (`compute_blob_kzg_proof`, `verify_blob_kzg_proof_batch`)
are reported under "extra" and are not part of `value`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_BLOB = 131072
ALG_BYTES_COMMIT = 131072 + 48  # blob in + commitment out
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
SEED = 0x4844


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="blobs per GPU per step (BASELINE configs[1]: 4096)")
    ap.add_argument("--window-bits", type=int, default=int(os.environ.get("KATETH_AMD_WINDOW_BITS", "16")),
                    help="fixed-base window c (table: c=16 -> 192 GiB of the 288 GB HBM, 16 s to build; c=15 -> 102 GiB, 9 s; c=12 -> 16 GiB, 1 s); falls back to smaller windows if the table cannot be allocated")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, the measured path) or gloo (rehearsal: gathers via host)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary proof/verify workloads")
    ap.add_argument("--cpu-sample", type=int, default=0, help="blobs in the CPU baseline sample (0 = auto, ~10-30 s)")
    return ap.parse_args()


def cpu_baseline(sample_blobs, setup_path, gpu_out48):
    """oracle/cport (C port of kateth's CPU path: Pippenger c=10 signed digits as
    blst uses, bases re-normalised on every call, one thread and all host cores)
    timed on a bounded sample of the same synthetic blobs, rank 0 only.  Checker
    code: never the thing measured as `value`.  Its outputs are compared byte for
    byte with the GPU's for the same blobs."""
    from oracle.cport import binding

    res = binding.time_commitment(None, setup_path, sample_blobs, SEED)
    raw = res.pop("_raw_outputs")
    res.pop("outputs", None)
    res["matches_gpu_bytes"] = bool(raw == gpu_out48[: len(raw)])
    return res


def pmc_traffic(n, window_bits):
    """HBM bytes per MSM-kernel launch from the committed rocprofv3 PMC passes
    (profiles/r01/pmc_traffic.json; separate --pmc FETCH_SIZE / WRITE_SIZE runs of this
    same command, calibrated against the kernel's known gather bytes as MI355X_MICROARCH.md
    prescribes: factor 1.0 for this access pattern, see the note in that file).
    bench.py cannot run the profiler on itself, so this is the profiled value for the
    same (batch, window) configuration, or null when none has been recorded."""
    path = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
    try:
        rec = json.load(open(path))
        key = "n%d_c%d" % (n, window_bits)
        return rec[key]["hbm_bytes_per_launch"] if key in rec else None
    except (OSError, ValueError, KeyError):
        return None


def extra_workloads(torch, setup, dev, stream, d_blobs, d_commitments, n):
    """BASELINE.json configs[2] and configs[3] (secondary numbers, not `value`):
    compute_blob_kzg_proof on the same 4096 resident blobs, and
    verify_blob_kzg_proof_batch on 65,536 resident (blob, commitment, proof) triples
    (the 4096 blobs tiled 16x: every triple is valid, so the batch verifies true)."""
    import time as _t

    out = {}
    d_proofs = torch.empty(n * 48, dtype=torch.uint8, device=dev)
    d_status = torch.empty(n, dtype=torch.int32, device=dev)
    setup.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_commitments.data_ptr(), n, d_proofs.data_ptr(), d_status.data_ptr(), stream)
    torch.cuda.synchronize()
    assert int(d_status.abs().sum()) == 0
    t0 = _t.perf_counter()
    reps = 2
    for _ in range(reps):
        setup.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_commitments.data_ptr(), n, d_proofs.data_ptr(), d_status.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = (_t.perf_counter() - t0) / reps
    out["compute_blob_kzg_proof"] = {"workload": "batch=%d blobs resident in HBM" % n, "blobs_per_s": n / dt, "ms_per_batch": 1e3 * dt,
                                     "algorithmic_GBps": n * 131168 / dt / 1e9}
    tile = max(1, 65536 // n)
    nv = tile * n
    vb = d_blobs.repeat(tile)
    vc = d_commitments.repeat(tile)
    vp = d_proofs.repeat(tile)
    torch.cuda.synchronize()
    ok = setup.verify_blob_proof_batch_dev(vb.data_ptr(), vc.data_ptr(), vp.data_ptr(), nv, stream)
    assert ok is True, "verify_blob_kzg_proof_batch must accept the engine's own proofs"
    t0 = _t.perf_counter()
    for _ in range(reps):
        ok = setup.verify_blob_proof_batch_dev(vb.data_ptr(), vc.data_ptr(), vp.data_ptr(), nv, stream)
    dt = (_t.perf_counter() - t0) / reps
    out["verify_blob_kzg_proof_batch"] = {"workload": "batch=%d (blob, commitment, proof) triples resident in HBM, includes the host pairing" % nv,
                                          "blobs_per_s": nv / dt, "ms_per_batch": 1e3 * dt, "result": bool(ok),
                                          "algorithmic_GBps": nv * 131168 / dt / 1e9, "hbm_frac_of_8TBps": nv * 131168 / dt / 8e12}
    # CPU baseline for the same metric: C port of the reference's verify path on the first 32 triples
    try:
        from oracle.cport import binding

        m = min(32, n)
        hb = d_blobs[: m * BYTES_PER_BLOB].cpu().numpy().tobytes()
        hc = d_commitments[: m * 48].cpu().numpy().tobytes()
        hp = d_proofs[: m * 48].cpu().numpy().tobytes()
        out["verify_blob_kzg_proof_batch"]["cpu_baseline"] = binding.time_verify(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), hb, hc, hp, m)
    except Exception as err:  # noqa: BLE001
        out["verify_blob_kzg_proof_batch"]["cpu_baseline"] = {"value": None, "error": repr(err)}
    # single-item latencies (BASELINE configs[0] shape: one blob, as benches/kzg.rs:35-43 times them)
    lat = {}
    for name, fn in (
        ("blob_to_kzg_commitment", lambda: setup.blob_to_commitment_batch_dev(d_blobs.data_ptr(), 1, d_commitments.data_ptr(), d_status.data_ptr(), stream)),
        ("compute_blob_kzg_proof", lambda: setup.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_commitments.data_ptr(), 1, d_proofs.data_ptr(), d_status.data_ptr(), stream)),
        ("verify_blob_kzg_proof", lambda: setup.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_commitments.data_ptr(), d_proofs.data_ptr(), 1, stream)),
    ):
        fn()
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        for _ in range(5):
            fn()
            torch.cuda.synchronize()
        lat[name] = 1e3 * (_t.perf_counter() - t0) / 5
    out["single_blob_latency_ms"] = lat
    # a corrupted proof must flip the result
    vp[48 * 7:48 * 8] = vp[0:48]
    torch.cuda.synchronize()
    out["verify_blob_kzg_proof_batch"]["rejects_corrupted_batch"] = (setup.verify_blob_proof_batch_dev(vb.data_ptr(), vc.data_ptr(), vp.data_ptr(), nv, stream) is False)
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the engine has no CPU fallback"
    ndev = torch.cuda.device_count()
    local_dev = local_rank % max(1, ndev)  # one GPU per rank on a real node; ranks share a card only in the gloo rehearsal
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)

    import kateth_amd

    setup_path = os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")
    t0 = time.time()
    setup = None
    tried = []
    for c in [args.window_bits] + [w for w in (15, 14, 12) if w < args.window_bits]:
        try:  # the table is sized for 288 GB of HBM; step down if this device cannot hold it
            setup = kateth_amd.Setup.load_json(setup_path, device=local_dev, window_bits=c)
            break
        except kateth_amd.kzg.EngineError as err:
            tried.append("c=%d: %s" % (c, err))
            torch.cuda.empty_cache()
    assert setup is not None, "context creation failed for every window size: %r" % tried
    t_setup = time.time() - t0

    n = args.batch
    d_blobs = torch.empty(n * BYTES_PER_BLOB, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n * 48, dtype=torch.uint8, device=dev)
    d_status = torch.empty(n, dtype=torch.int32, device=dev)
    gathered = torch.empty(world * n * 48, dtype=torch.uint8, device=dev) if world > 1 else None
    stream = torch.cuda.current_stream().cuda_stream
    setup.synth_blobs_dev(SEED, rank * n, n, d_blobs.data_ptr(), stream)

    def step():
        setup.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_out.data_ptr(), d_status.data_ptr(), stream)
        if world > 1:
            if args.backend == "nccl":
                dist.all_gather_into_tensor(gathered, d_out)  # RCCL over xGMI: 48 B per blob
            else:  # gloo rehearsal path: stage through the host
                host = [torch.empty(n * 48, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(host, d_out.cpu())
                gathered.copy_(torch.cat(host))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    setup.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = setup.profile_end()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert int(d_status.abs().sum()) == 0, "synthetic blobs must all be valid"

    # ---- correctness spot check of the timed output against the oracle golden vectors
    gpu_out = d_out.cpu().numpy().tobytes() if rank == 0 else b""
    if rank == 0:
        golden = json.load(open(os.path.join(ROOT, "tests", "golden", "kzg_vectors.json")))
        out = gpu_out
        for rec in golden["blobs"]:
            b = rec["index"]
            if b < n:
                assert out[48 * b:48 * b + 48].hex() == rec["commitment"], "GPU commitment != oracle golden vector"

    blobs_per_s = world * n * args.steps / elapsed
    result = {
        "metric": "blobs/sec for blob_to_kzg_commitment (n=4096 field elements per blob)",
        "value": blobs_per_s,
        "unit": "blobs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 limbs (381-bit Fp / 255-bit Fr Montgomery integer arithmetic; 28/29-bit radix in the hot loops)",
        "data": "synthetic: element(b,i)=SHA-256(seed||b||i) mod r, generated on device, resident in HBM",
        "config": {
            "workload": "blob_to_kzg_commitment batch=%d blobs per GPU (BASELINE configs[1])" % n,
            "blobs_per_gpu": n,
            "window_bits": setup.window_bits,
            "table_gib": setup.table_bytes / 2**30,
            "parallelism": "blob-sharded x%d, RCCL all-gather of 48-B commitments" % world,
            "setup_s": t_setup,
        },
    }
    if rank == 0:
        # measured integer-ALU ceiling: dependent Fp Montgomery multiplies with the multiply of the MSM kernel in
        # use (radix-2^28 limbs by default), 8 waves/SIMD, whole chip.  A mixed add is 10 products (2 of them
        # squarings, 2 sharing one reduction) plus ~900 other VALU instructions, so valu_frac is a utilisation
        # estimate, not an exact instruction ratio; DESIGN.md section 5 gives the instruction counts.
        lanes = 256 * 4 * 64 * 8
        setup.microbench_fp_mul(lanes, 200)
        prof["fp_mul_peak_per_s"] = lanes * 2000 / (setup.microbench_fp_mul(lanes, 2000) * 1e-3)
        k_ms = prof["msm_ms"] / max(1, prof["msm_launches"])
        alg_bytes = ALG_BYTES_COMMIT * n  # per launch: one launch processes the rank's whole batch
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None
        adds_per_blob = prof["adds_per_blob"]
        result["roofline"] = {
            "kernel": "k_msm_fixed" if os.environ.get("KATETH_AMD_MSM_RADIX") == "32" else "k_msm_fixed28",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
            "traffic": pmc_traffic(n, setup.window_bits),
            "kernel_ms": k_ms,
            "launches": prof["msm_launches"],
            "algorithmic_bytes_per_blob": ALG_BYTES_COMMIT,
            "note": "integer-ALU bound, not HBM bound: see valu_* fields and DESIGN.md section 5",
            "table_gather_bytes_per_blob": adds_per_blob * 96,
            # multiply-equivalents per mixed add, weighted by v_mad_u64_u32 count: radix-2^28 kernel 6 products + 2 squarings
            # (301/392 each) + 2 products sharing one reduction (588/392) = 9.04; 32-bit-limb kernel 10
            "valu_fp_mul_per_s": (adds_per_blob * (10.0 if os.environ.get("KATETH_AMD_MSM_RADIX") == "32" else 9.04) * n / (k_ms * 1e-3)) if k_ms > 0 else None,
            "valu_fp_mul_peak_per_s": prof.get("fp_mul_peak_per_s"),
        }
        if result["roofline"]["valu_fp_mul_peak_per_s"]:
            result["roofline"]["valu_frac"] = result["roofline"]["valu_fp_mul_per_s"] / result["roofline"]["valu_fp_mul_peak_per_s"]
        if not args.no_extra and world == 1:
            try:
                result["extra"] = extra_workloads(torch, setup, dev, stream, d_blobs, d_out, n)
            except Exception as err:  # secondary numbers never hide the headline
                result["extra"] = {"error": repr(err)}
        if not args.no_cpu_baseline and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline(args.cpu_sample, setup_path, gpu_out)
            except Exception as err:  # the baseline is reporting only; never hide the GPU number
                result["cpu_baseline"] = {"value": None, "error": repr(err)}
        print(json.dumps(result), flush=True)
    setup.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
