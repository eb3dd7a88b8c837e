#!/usr/bin/env python3
"""What the GROUP protocol costs, separated from what two members sharing ONE card cost each other (every box here has one GPU):
verify_blob_kzg_proof_batch of 2 x 32,768 device-resident triples
  (a) through a two-member group context's kzg_verify_blob_proof_batch_group_dev (phase 1 / roots / phase 2 / one pairing),
  (b) as two INDEPENDENT single-device calls of 32,768 triples on two contexts from two host threads (same contention on the
      card, no protocol: two pairings, no exchange),
  (c) as one single-device call of 65,536 triples (what one member would do alone).
(a) - (b) is the protocol; (b) - (c) is the shared card (the 32,768-triple hash is the two-wave kernel: two of them at once put
two hash waves on every SIMD).  Prints one JSON object."""
import concurrent.futures
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
wb = 16
one = kateth_amd.Setup.load_json(SETUP, window_bits=wb)
two = kateth_amd.Setup.load_json(SETUP, window_bits=wb)
group = kateth_amd.Setup.load_json(SETUP, window_bits=wb, devices=[0, 0])
N = 2 * n
d_b = torch.empty(N * 131072, dtype=torch.uint8, device="cuda")
d_c = torch.empty(N * 48, dtype=torch.uint8, device="cuda")
d_p = torch.empty(N * 48, dtype=torch.uint8, device="cuda")
d_s = torch.empty(N, dtype=torch.int32, device="cuda")
one.synth_blobs_dev(0x4844, 0, N, d_b.data_ptr())
one.blob_to_commitment_batch_dev(d_b.data_ptr(), N, d_c.data_ptr(), d_s.data_ptr())
one.compute_blob_proof_batch_dev(d_b.data_ptr(), d_c.data_ptr(), N, d_p.data_ptr(), d_s.data_ptr())
torch.cuda.synchronize()
assert int(d_s.abs().sum()) == 0
halves = [(d_b[k * n * 131072:].data_ptr(), d_c[k * n * 48:].data_ptr(), d_p[k * n * 48:].data_ptr()) for k in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]


def timeit(fn, reps=8):
    assert fn() is True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        assert fn() is True
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def via_group():
    return group.verify_blob_proof_batch_group_dev([h[0] for h in halves], [h[1] for h in halves], [h[2] for h in halves], [n, n], [s.cuda_stream for s in streams])


pool = concurrent.futures.ThreadPoolExecutor(max_workers=2)


def independent():
    def half(k):
        torch.cuda.set_device(0)
        return (one, two)[k].verify_blob_proof_batch_dev(*halves[k], n, streams[k].cuda_stream)

    return all(pool.map(half, range(2)))


def single():
    return one.verify_blob_proof_batch_dev(d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), N)


out = {"triples_total": N, "table_class": wb,
       "a_group_of_two_members_one_card_ms": timeit(via_group), "b_two_independent_contexts_two_threads_ms": timeit(independent),
       "c_one_single_device_call_ms": timeit(single), "half_alone_ms": timeit(lambda: one.verify_blob_proof_batch_dev(*halves[0], n))}
out["protocol_overhead_ms"] = out["a_group_of_two_members_one_card_ms"] - out["b_two_independent_contexts_two_threads_ms"]
out["note"] = ("on a node with one GPU per member every member runs `half_alone_ms` of kernels on a card of its own; the group call then costs that + the "
               "protocol overhead (two joins of pooled host threads, the exchange of 32-byte roots in host memory, one shared-squaring pairing instead of the "
               "two parallel Miller loops of the single-device ending)")
print(json.dumps(out))
for s_ in (group, two, one):
    s_.close()
