#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) throughput of the batch entry points: the caller's blobs live in host memory, as in
kateth's byte-slice API; every call allocates, copies in, computes, copies out."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c = int(sys.argv[2]) if len(sys.argv) > 2 else 12
s = kateth_amd.Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), window_bits=c)
d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
d_st = torch.empty(n, dtype=torch.int32, device="cuda")
s.synth_blobs_dev(0x4844, 0, n, d_blobs.data_ptr())
s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
torch.cuda.synchronize()
blobs = d_blobs.cpu().numpy().tobytes()
cs, ps = d_c.cpu().numpy().tobytes(), d_p.cpu().numpy().tobytes()
out = {"n": n, "window_bits": c}


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


t = timed(lambda: s.blob_to_commitment_batch(blobs, n))
out["commit_host_ms"], out["commit_host_blobs_per_s"] = 1e3 * t, n / t
t = timed(lambda: (s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr()), torch.cuda.synchronize()))
out["commit_dev_ms"], out["commit_dev_blobs_per_s"] = 1e3 * t, n / t
got, st = s.blob_to_commitment_batch(blobs, n)
assert got == cs and not any(st)
t = timed(lambda: s.compute_blob_proof_batch(blobs, cs))
out["proof_host_ms"], out["proof_host_blobs_per_s"] = 1e3 * t, n / t
bl = [blobs[i * 131072:(i + 1) * 131072] for i in range(n)]
cl = [cs[i * 48:(i + 1) * 48] for i in range(n)]
pl = [ps[i * 48:(i + 1) * 48] for i in range(n)]
t = timed(lambda: s.verify_blob_proof_batch(bl, cl, pl))
out["verify_host_ms"], out["verify_host_blobs_per_s"] = 1e3 * t, n / t
t = timed(lambda: s.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n))
out["verify_dev_ms"], out["verify_dev_blobs_per_s"] = 1e3 * t, n / t
print(json.dumps(out))
s.close()
