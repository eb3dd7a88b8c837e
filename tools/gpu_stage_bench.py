#!/usr/bin/env python3
"""Times compute_blob_kzg_proof (n_proof blobs) and verify_blob_kzg_proof_batch (n_verify blobs) end to end,
inputs resident in HBM.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
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
    n_proof = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n_verify = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    c = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    s = kateth_amd.Setup.load_json(SETUP, window_bits=c)
    n = max(n_proof, n_verify)
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_status = torch.empty(n, dtype=torch.int32, device="cuda")
    s.synth_blobs_dev(0x4844, 0, n, d_blobs.data_ptr())
    t0 = time.perf_counter()
    s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_status.data_ptr())
    torch.cuda.synchronize()
    t_commit = time.perf_counter() - t0
    t0 = time.perf_counter()
    s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_status.data_ptr())
    torch.cuda.synchronize()
    t_proof_all = time.perf_counter() - t0
    assert int(d_status.abs().sum()) == 0
    out = {"c": c, "n": n, "commit_blobs_per_s": n / t_commit, "proof_blobs_per_s_all": n / t_proof_all}
    # timed proof batch
    reps = 2
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n_proof, d_p.data_ptr(), d_status.data_ptr())
    torch.cuda.synchronize()
    out["proof_n"] = n_proof
    out["proof_blobs_per_s"] = n_proof * reps / (time.perf_counter() - t0)
    # timed verify batch
    ok = s.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n_verify)
    assert ok is True
    t0 = time.perf_counter()
    for _ in range(reps):
        ok = s.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n_verify)
    dt = (time.perf_counter() - t0) / reps
    out["verify_n"] = n_verify
    out["verify_s"] = dt
    out["verify_blobs_per_s"] = n_verify / dt
    for small in (1, 8, 128):
        t0 = time.perf_counter()
        ok = s.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), small)
        out["verify_ms_n%d" % small] = 1e3 * (time.perf_counter() - t0)
    print(json.dumps(out), flush=True)
    s.close()


if __name__ == "__main__":
    main()
