import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, kateth_amd
s = kateth_amd.Setup.load_json("tests/golden/trusted_setup_4096.json", window_bits=8)
n = 65536
d_b = torch.empty(n*131072, dtype=torch.uint8, device="cuda"); d_c = torch.empty(n*48, dtype=torch.uint8, device="cuda"); d_p = torch.empty(n*48, dtype=torch.uint8, device="cuda"); d_s = torch.empty(n, dtype=torch.int32, device="cuda")
s.synth_blobs_dev(1, 0, n, d_b.data_ptr()); s.blob_to_commitment_batch_dev(d_b.data_ptr(), n, d_c.data_ptr(), d_s.data_ptr()); s.compute_blob_proof_batch_dev(d_b.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_s.data_ptr()); torch.cuda.synchronize()
for _ in range(3):
    t0=time.perf_counter(); ok = s.verify_blob_proof_batch_dev(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), n); print("verify", ok, 1e3*(time.perf_counter()-t0), file=sys.stderr)
