#!/usr/bin/env python3
"""compute_blob_kzg_proof throughput on resident blobs (env KATETH_AMD_PROOF_CHUNK / _OVERLAP select the pipeline shape)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c = int(sys.argv[2]) if len(sys.argv) > 2 else 12
s = kateth_amd.Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), window_bits=c, table_max=True)
d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
d_st = torch.empty(n, dtype=torch.int32, device="cuda")
s.synth_blobs_dev(0x4844, 0, n, d_blobs.data_ptr())
s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
VARIANTS = (("chunk4096_serial", ("4096", "0")), ("chunk4096_overlap", ("4096", "1")), ("chunk2048_overlap", ("2048", "1")), ("chunk1024_overlap", ("1024", "1")), ("chunk8192_overlap", ("8192", "1")))
for tag, env in VARIANTS:
    os.environ["KATETH_AMD_PROOF_CHUNK"], os.environ["KATETH_AMD_PROOF_OVERLAP"] = env
    s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-20s %.1f ms  %.0f blobs/s" % (tag, dt * 1e3, n / dt), flush=True)
s.close()
