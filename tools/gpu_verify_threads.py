#!/usr/bin/env python3
"""verify_blob_kzg_proof_batch at 65,536 triples with T calls in flight from T host threads (a call is synchronous: it returns
the boolean), T = 1..4, each thread on a stream of its own; prints one JSON line.   usage: gpu_verify_threads.py [calls=12]"""
import concurrent.futures
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
s = kateth_amd.Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), window_bits=0, table_max=True)
n = 65536
vb = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
vc = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
vp = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
st = torch.empty(n, dtype=torch.int32, device="cuda")
s.synth_blobs_dev(0x7EE7, 0, n, vb.data_ptr())
s.blob_to_commitment_batch_dev(vb.data_ptr(), n, vc.data_ptr(), st.data_ptr())
s.compute_blob_proof_batch_dev(vb.data_ptr(), vc.data_ptr(), n, vp.data_ptr(), st.data_ptr())
torch.cuda.synchronize()
assert int(st.abs().sum()) == 0
out = {"n": n, "calls_per_measurement": calls, "table_class": s.window_bits, "threads": {}}
for T in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(T)]

    def one(k):
        torch.cuda.set_device(0)
        return s.verify_blob_proof_batch_dev(vb.data_ptr(), vc.data_ptr(), vp.data_ptr(), n, streams[k % T].cuda_stream)

    with concurrent.futures.ThreadPoolExecutor(max_workers=T) as pool:
        assert all(pool.map(one, range(2 * T)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        oks = list(pool.map(one, range(calls)))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / calls
    assert all(v is True for v in oks)
    out["threads"][str(T)] = {"ms_per_call": 1e3 * dt, "blobs_per_s": n / dt}
print(json.dumps(out))
