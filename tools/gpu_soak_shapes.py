#!/usr/bin/env python3
"""Soak of the launch-shape chooser and of the host-buffer pipelines (round 3): many batch sizes around the places where
msm_shape changes its answer (multiples of 1,024 / 2,048 / 4,096 waves, +-1..3; the latency comb's 4 / 8 / 16 limits; the host
commit ramp's 768 / 4,096 / 8,192 limits) through
  * blob_to_kzg_commitment, device-resident, default context (class 22) against the class-8 engine;
  * the same sizes through the HOST-buffer entry point (ramped chunk pipeline on two streams);
  * compute_blob_kzg_proof (device and host) on a subset.
Every output byte is compared.  usage: gpu_soak_shapes.py [rounds=1]   (JSON summary on stdout)"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NMAX = 16500
big = kateth_amd.Setup.load_json(SETUP, window_bits=0, table_max=True)
ref = kateth_amd.Setup.load_json(SETUP, window_bits=8)
d_blobs = torch.empty(NMAX * 131072, dtype=torch.uint8, device="cuda")
big.synth_blobs_dev(0x50A4, 0, NMAX, d_blobs.data_ptr())
want_c = torch.empty(NMAX * 48, dtype=torch.uint8, device="cuda")
want_p = torch.empty(NMAX * 48, dtype=torch.uint8, device="cuda")
st = torch.empty(NMAX, dtype=torch.int32, device="cuda")
for first in range(0, NMAX, 1000):  # the reference results: class 8, ragged small batches
    m = min(1000, NMAX - first)
    ref.blob_to_commitment_batch_dev(d_blobs.data_ptr() + first * 131072, m, want_c.data_ptr() + first * 48, st.data_ptr() + first * 4)
    ref.compute_blob_proof_batch_dev(d_blobs.data_ptr() + first * 131072, want_c.data_ptr() + first * 48, m, want_p.data_ptr() + first * 48,
                                     st.data_ptr() + first * 4)
torch.cuda.synchronize()
assert int(st.abs().sum()) == 0
wc, wp = want_c.cpu().numpy().tobytes(), want_p.cpu().numpy().tobytes()
host_blobs = d_blobs.cpu().numpy()

sizes = set()
for base in (1, 4, 8, 16, 64, 512, 768, 1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384):
    for d in (-3, -1, 0, 1, 2, 5):
        if 1 <= base + d <= NMAX:
            sizes.add(base + d)
rng = random.Random(0xC0FFEE)
sizes |= {rng.randrange(1, NMAX) for _ in range(24)}
sizes = sorted(sizes)
stats = {"sizes": len(sizes), "rounds": rounds, "commit_dev": 0, "commit_host": 0, "proof_dev": 0, "proof_host": 0, "mismatches": 0, "blobs_checked": 0}
t0 = time.time()
for r in range(rounds):
    for n in sizes:
        off = rng.randrange(0, NMAX - n + 1)  # a different window of the blobs every time
        c = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
        s2 = torch.full((n,), -5, dtype=torch.int32, device="cuda")
        big.blob_to_commitment_batch_dev(d_blobs.data_ptr() + off * 131072, n, c.data_ptr(), s2.data_ptr())
        torch.cuda.synchronize()
        ok = int(s2.abs().sum()) == 0 and c.cpu().numpy().tobytes() == wc[48 * off:48 * (off + n)]
        stats["commit_dev"] += 1
        got, gst = big.blob_to_commitment_batch(host_blobs[off * 131072:(off + n) * 131072].tobytes(), n)
        ok2 = not any(gst) and got == wc[48 * off:48 * (off + n)]
        stats["commit_host"] += 1
        ok3 = ok4 = True
        if n <= 4100 or n in (8192, 16384):
            p = torch.zeros(n * 48, dtype=torch.uint8, device="cuda")
            big.compute_blob_proof_batch_dev(d_blobs.data_ptr() + off * 131072, want_c.data_ptr() + off * 48, n, p.data_ptr(), s2.data_ptr())
            torch.cuda.synchronize()
            ok3 = int(s2.abs().sum()) == 0 and p.cpu().numpy().tobytes() == wp[48 * off:48 * (off + n)]
            stats["proof_dev"] += 1
            if n <= 2100:
                gp, gpst = big.compute_blob_proof_batch(host_blobs[off * 131072:(off + n) * 131072].tobytes(), wc[48 * off:48 * (off + n)])
                ok4 = not any(gpst) and gp == wp[48 * off:48 * (off + n)]
                stats["proof_host"] += 1
        stats["blobs_checked"] += n
        if not (ok and ok2 and ok3 and ok4):
            stats["mismatches"] += 1
            print("MISMATCH n=%d off=%d dev=%s host=%s proof=%s proof_host=%s" % (n, off, ok, ok2, ok3, ok4), file=sys.stderr, flush=True)
    print("round %d done: %d sizes, %.0f s" % (r, len(sizes), time.time() - t0), file=sys.stderr, flush=True)
stats["seconds"] = time.time() - t0
stats["table_class"], stats["plane_groups"] = big.window_bits, big.plane_groups
print(json.dumps(stats))
big.close()
ref.close()
sys.exit(1 if stats["mismatches"] else 0)
