#!/usr/bin/env python3
"""GPU probe: Fp-multiply throughput vs occupancy, and commitment throughput vs window size."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")


def main():
    windows = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,12").split(",")]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    out = {"fp_mul": [], "msm": []}
    first = True
    for c in windows:
        t0 = time.time()
        s = kateth_amd.Setup.load_json(SETUP, window_bits=c)
        t_setup = time.time() - t0
        if first:
            first = False
            for waves_per_simd in (1, 2, 3, 4, 6, 8):
                lanes = 256 * 4 * 64 * waves_per_simd
                iters = 2000
                s.microbench_fp_mul(lanes, 200)
                ms = s.microbench_fp_mul(lanes, iters)
                rate = lanes * iters / (ms * 1e-3)
                out["fp_mul"].append({"waves_per_simd": waves_per_simd, "ms": ms, "fp_mul_per_s": rate})
                print("fp_mul waves/SIMD=%d  %.3f ms  %.2f G mul/s" % (waves_per_simd, ms, rate / 1e9), flush=True)
        d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
        d_out = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        d_status = torch.empty(n, dtype=torch.int32, device="cuda")
        s.synth_blobs_dev(0x4844, 0, n, d_blobs.data_ptr())
        s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_out.data_ptr(), d_status.data_ptr())
        torch.cuda.synchronize()
        s.profile_begin()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_out.data_ptr(), d_status.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        prof = s.profile_end()
        rec = {"c": c, "table_gib": s.table_bytes / 2**30, "setup_s": t_setup, "n": n, "s_per_batch": dt, "blobs_per_s": n / dt,
               "kernel_ms": prof["msm_ms"] / prof["msm_launches"], "adds_per_blob": prof["adds_per_blob"],
               "madd_per_s": prof["adds_per_blob"] * n / (prof["msm_ms"] / prof["msm_launches"] * 1e-3)}
        out["msm"].append(rec)
        print(json.dumps(rec), flush=True)
        s.close()
        del d_blobs
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
