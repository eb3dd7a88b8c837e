#!/usr/bin/env python3
"""The in-process GROUP context (kzg_config.devices / ndev, engine_multi.hip) at the benchmark's batch sizes on ONE card: the
device ordinal listed twice = two members, each with a complete class-16 table, sharing the GPU.  No speed-up can come of it here
(one chip) -- the point is that the sharded host-buffer entry points run the metric's batches (4,096 blobs; verification of 16,384
triples) through two members side by side, bit-exact against the single-device context, and what the sharding itself costs.
Prints one JSON line.   usage: gpu_group_bench.py [n=4096] [n_verify=16384]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import kateth_amd  # noqa: E402

SETUP = os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
one = kateth_amd.Setup.load_json(SETUP, window_bits=16)
grp = kateth_amd.Setup.load_json(SETUP, window_bits=16, devices=[0, 0])
m = max(n, nv)
d_blobs = torch.empty(m * 131072, dtype=torch.uint8, device="cuda")
d_c = torch.empty(m * 48, dtype=torch.uint8, device="cuda")
d_p = torch.empty(m * 48, dtype=torch.uint8, device="cuda")
d_st = torch.empty(m, dtype=torch.int32, device="cuda")
one.synth_blobs_dev(0x4844, 0, m, d_blobs.data_ptr())
one.blob_to_commitment_batch_dev(d_blobs.data_ptr(), m, d_c.data_ptr(), d_st.data_ptr())
torch.cuda.synchronize()
one.compute_blob_proof_batch_dev(d_blobs.data_ptr(), d_c.data_ptr(), m, d_p.data_ptr(), d_st.data_ptr())
torch.cuda.synchronize()
assert int(d_st.abs().sum()) == 0
hb, hc, hp = d_blobs.cpu().numpy().tobytes(), d_c.cpu().numpy().tobytes(), d_p.cpu().numpy().tobytes()
del d_blobs
torch.cuda.empty_cache()
out = {"members": grp.members, "member_devices": [grp.member_device(k) for k in range(grp.members)], "table_class": grp.window_bits, "n": n, "n_verify": nv,
       "note": "two members on ONE card: functional evidence at batch size, not a scaling measurement"}


hb_n, hc_n, hb_v, hc_v, hp_v = hb[: n * 131072], hc[: 48 * n], hb[: nv * 131072], hc[: 48 * nv], hp[: 48 * nv]  # sliced ONCE (a slice of bytes is a copy)
bad_v = hp[48:48 * nv] + hp[:48]


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps, r


for name, ctx in (("single", one), ("group", grp)):
    t, (c, st) = timed(lambda: ctx.blob_to_commitment_batch(hb_n, n))
    assert c == hc[: 48 * n] and not any(st), name
    out["commit_%s_blobs_per_s" % name] = n / t
    t, (p, st) = timed(lambda: ctx.compute_blob_proof_batch(hb_n, hc_n))
    assert p == hp[: 48 * n] and not any(st), name
    out["proof_%s_blobs_per_s" % name] = n / t
    t, ok = timed(lambda: ctx.verify_blob_proof_batch_host(hb_v, hc_v, hp_v, nv))
    assert ok is True, name
    out["verify_%s_blobs_per_s" % name] = nv / t
    assert ctx.verify_blob_proof_batch_host(hb_v, hc_v, bad_v, nv) is False, name
out["bit_exact"] = True
print(json.dumps(out))
grp.close()
one.close()
