#!/usr/bin/env python3
"""Soak of calls kept in flight (workspace slots, round 4): `rounds` rounds of commitment / proof calls of random sizes on two or
three streams with no synchronisation inside a round, the table swapped in by the background build under way in the first
rounds (KZG_CFG_BUILD_ASYNC, class 16), every output compared with a class-8 context's one-call-at-a-time results; and batch
verifications from two host threads beside them.  Prints one JSON line.   usage: gpu_soak_inflight.py [rounds=40]"""
import json
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(0x50AC)
N = 2200
ref = kateth_amd.Setup.load_json(SETUP, window_bits=8)
d_blobs = torch.empty(N * 131072, dtype=torch.uint8, device="cuda")
d_c = torch.empty(N * 48, dtype=torch.uint8, device="cuda")
d_p = torch.empty(N * 48, dtype=torch.uint8, device="cuda")
d_st = torch.empty(N, dtype=torch.int32, device="cuda")
ref.synth_blobs_dev(0x50AC, 0, N, d_blobs.data_ptr())
ref.blob_to_commitment_batch_dev(d_blobs.data_ptr(), N, d_c.data_ptr(), d_st.data_ptr())
torch.cuda.synchronize()
ref.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), N, d_p.data_ptr(), d_st.data_ptr())
torch.cuda.synchronize()
assert int(d_st.abs().sum()) == 0
t0 = time.time()
eng = kateth_amd.Setup.load_json(SETUP, window_bits=16, build_async=True)
create_s = time.time() - t0
streams = [torch.cuda.Stream() for _ in range(3)]
stats = {"rounds": rounds, "calls": 0, "mismatches": 0, "verify_calls": 0, "verify_wrong": 0, "classes_seen": [], "create_s": create_s}
stop = threading.Event()


def verifier(tid):
    st = torch.cuda.Stream()
    while not stop.is_set():
        m = rng.choice([1, 7, 64, 300, 1025])
        off = rng.randrange(0, N - m)
        ok = eng.verify_blob_proof_batch_dev(d_blobs.data_ptr() + off * 131072, d_c.data_ptr() + off * 48, d_p.data_ptr() + off * 48, m, st.cuda_stream)
        bad = eng.verify_blob_proof_batch_dev(d_blobs.data_ptr() + off * 131072, d_c.data_ptr() + off * 48, d_p.data_ptr() + (off + 1) * 48, m, st.cuda_stream)
        stats["verify_calls"] += 2
        if ok is not True or bad is not False:
            stats["verify_wrong"] += 1


threads = [threading.Thread(target=verifier, args=(t,)) for t in range(2)]
for t in threads:
    t.start()
for r in range(rounds):
    cls = eng.window_bits
    if cls not in stats["classes_seen"]:
        stats["classes_seen"].append(cls)
    outs = []
    for k in range(rng.randrange(3, 9)):
        m = rng.choice([1, 2, 16, 17, 100, 511, 512, 1000, 2048, 2200])
        st = streams[rng.randrange(3)]
        with torch.cuda.stream(st):  # the fills run on the call's own stream (torch's pool streams do not order against its default stream)
            o = torch.zeros(m * 48, dtype=torch.uint8, device="cuda")
            s = torch.full((m,), -3, dtype=torch.int32, device="cuda")
            if rng.random() < 0.5:
                eng.blob_to_commitment_batch_dev(d_blobs.data_ptr(), m, o.data_ptr(), s.data_ptr(), st.cuda_stream)
                outs.append((o, s, d_c[: 48 * m]))
            else:
                eng.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), m, o.data_ptr(), s.data_ptr(), st.cuda_stream)
                outs.append((o, s, d_p[: 48 * m]))
    torch.cuda.synchronize()
    for o, s, want in outs:
        stats["calls"] += 1
        if int(s.abs().sum()) != 0 or not torch.equal(o, want):
            stats["mismatches"] += 1
eng.wait_ready()
stats["final_class"] = eng.window_bits
stop.set()
for t in threads:
    t.join(timeout=120)
stats["seconds"] = time.time() - t0
print(json.dumps(stats))
eng.close()
ref.close()
sys.exit(1 if stats["mismatches"] or stats["verify_wrong"] else 0)
