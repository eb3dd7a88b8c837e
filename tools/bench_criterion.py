#!/usr/bin/env python3
"""Criterion-equivalent of the reference's benches/kzg.rs:10-65 through the host mirror of kateth's API.

Same shape as the reference bench: one blob through blob_to_commitment / blob_proof / verify_blob_proof, then
verify_blob_proof_batch for batch sizes 1, 2, 4, ..., 128 with per-element throughput.  Every call takes HOST byte
buffers, as kateth's API does, so each timed call includes its own device allocation and PCIe copies; the
device-resident rate of the same batch is printed beside it.  The C port of the reference's CPU path (oracle/cport,
test infrastructure) is timed on the same inputs as the CPU column -- it is not kateth itself.

Prints one JSON object; commit it under profiles/.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")
SIZES = [1, 2, 4, 8, 16, 32, 64, 128]  # benches/kzg.rs:14
SEED = 0x4844


def timeit(fn, min_time=0.6, max_reps=200):
    fn()
    reps, t0 = 0, time.perf_counter()
    while True:
        fn()
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= min_time or reps >= max_reps:
            return dt / reps


def main():
    window_bits = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 0  # 0: the automatic choice (class 22 on an idle part)
    with_cpu = "--no-cpu" not in sys.argv
    s = kateth_amd.Setup.load_json(SETUP, window_bits=window_bits, table_max=True)
    n = SIZES[-1]
    d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
    d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    d_st = torch.empty(n, dtype=torch.int32, device="cuda")
    s.synth_blobs_dev(SEED, 0, n, d_blobs.data_ptr())
    s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
    s.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), n, d_p.data_ptr(), d_st.data_ptr())
    torch.cuda.synchronize()
    assert int(d_st.abs().sum()) == 0
    flat = d_blobs.cpu().numpy().tobytes()
    blobs = [flat[i * 131072:(i + 1) * 131072] for i in range(n)]
    cb, pb = d_c.cpu().numpy().tobytes(), d_p.cpu().numpy().tobytes()
    commitments = [cb[i * 48:(i + 1) * 48] for i in range(n)]
    proofs = [pb[i * 48:(i + 1) * 48] for i in range(n)]

    out = {"window_bits": s.window_bits, "table_gib": s.table_bytes / 2**30, "inputs": "synthetic blobs, seed 0x4844 (bench.py's generator)",
           "note": "host-buffer API calls (allocation + PCIe copies inside the timed call), as benches/kzg.rs times kateth's byte-slice API"}
    out["blob to kzg commitment"] = {"ms": 1e3 * timeit(lambda: s.blob_to_commitment(blobs[0]))}
    out["compute blob kzg proof"] = {"ms": 1e3 * timeit(lambda: s.blob_proof(blobs[0], commitments[0]))}
    out["verify blob kzg proof"] = {"ms": 1e3 * timeit(lambda: s.verify_blob_proof(blobs[0], commitments[0], proofs[0]))}
    assert s.verify_blob_proof(blobs[0], commitments[0], proofs[0]) is True
    group = {}
    for size in SIZES:
        t_host = timeit(lambda: s.verify_blob_proof_batch(blobs[:size], commitments[:size], proofs[:size]))
        t_dev = timeit(lambda: s.verify_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), size))
        group[str(size)] = {"ms": 1e3 * t_host, "elements_per_s": size / t_host, "device_resident_ms": 1e3 * t_dev,
                            "device_resident_elements_per_s": size / t_dev}
    assert s.verify_blob_proof_batch(blobs, commitments, proofs) is True
    out["verify blob kzg proof batch"] = group

    if with_cpu:
        from oracle.cport import binding  # test infrastructure: the timed CPU column only

        lib = binding.load()
        cs = binding.CSetup(lib, SETUP, subgroup_checks=False, threads=1)
        cpu = {}
        t, _ = cs.time_commitments(blobs[0], 1, 3)
        cpu["blob to kzg commitment"] = {"ms": 1e3 * t / 3 if t > 0 else None, "threads": 1}
        cs.set_threads(binding.host_cores())
        t, _ = cs.time_commitments(blobs[0], 1, 5)
        cpu["blob to kzg commitment (MSM tiled over host cores, as blst's pool does)"] = {"ms": 1e3 * t / 5, "threads": binding.host_cores()}
        cs.set_threads(1)
        grp = {}
        for size in SIZES:
            t = cs.time_verify_prepairing(b"".join(blobs[:size]), b"".join(commitments[:size]), b"".join(proofs[:size]), size, False)
            grp[str(size)] = {"ms": 1e3 * t, "elements_per_s": size / t}
        cpu["verify blob kzg proof batch (reference algorithm up to the pairing, 1 thread)"] = grp
        cpu["label"] = "CPU restatement of kateth/blst path (C port) -- not kateth itself"
        cs.close()
        out["cpu_port"] = cpu
    s.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
