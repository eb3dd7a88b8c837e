#!/usr/bin/env python3
"""Per-wave timing of one k_msm_comb30 launch (TEST-ONLY build, tests/window_msm/libkateth_amd_window_msm.so): where does
the difference between the kernel's duration and its waves' mean duration come from?  Prints, per XCD, the number of waves,
their mean/min/max duration and the shader clock they saw (cycles per microsecond of the 100 MHz wall clock).
usage: gpu_wave_times.py [n] [window_bits]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c = int(sys.argv[2]) if len(sys.argv) > 2 else 22
os.environ["KATETH_AMD_WAVE_TIMES"] = str(max(4 * n, 4096))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402
import kateth_amd  # noqa: E402

s = kateth_amd.Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), window_bits=c, lib_path=ge.build_test_engine())
d_blobs = torch.empty(n * 131072, dtype=torch.uint8, device="cuda")
d_c = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
d_st = torch.empty(n, dtype=torch.int32, device="cuda")
s.synth_blobs_dev(0x4844, 0, n, d_blobs.data_ptr())
for _ in range(3):
    s.blob_to_commitment_batch_dev(d_blobs.data_ptr(), n, d_c.data_ptr(), d_st.data_ptr())
torch.cuda.synchronize()
lib = s._lib
lib.kzg_test_read_wave_times.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
lib.kzg_test_read_wave_times.restype = ctypes.c_int32
out = {}
for units in sorted({n // 2, n, 2 * n, 4 * n} | {n * k for k in (3, 6, 8, 12, 16, 24, 32, 48, 64)}):
    if units == 0:
        continue
    buf = np.zeros((units, 4), dtype=np.uint64)
    if lib.kzg_test_read_wave_times(s._h, buf.ctypes.data, units) != 0:
        continue
    if not buf[:, 1].all():
        continue  # the launch had fewer units (the buffer is zeroed at context creation)
    t0 = buf[:, 0].min()
    start = (buf[:, 0] - t0) / 100.0  # microseconds
    end = (buf[:, 1] - t0) / 100.0
    dur = end - start
    mhz = buf[:, 2] / np.maximum(dur, 1e-9)
    xcc = (buf[:, 3] >> np.uint64(32)).astype(np.int64) & 0xF
    hw = buf[:, 3].astype(np.int64) & 0xFFFFFFFF
    rec = {"units": units, "kernel_span_us": float(end.max()), "wave_mean_us": float(dur.mean()), "wave_min_us": float(dur.min()), "wave_max_us": float(dur.max()),
           "start_max_us": float(start.max()), "shader_mhz_mean": float(mhz.mean()), "per_xcc": {}}
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        rec["per_xcc"][str(x)] = {"waves": int(m.sum()), "mean_us": float(dur[m].mean()), "max_us": float(dur[m].max()), "end_max_us": float(end[m].max()), "mhz": float(mhz[m].mean())}
    # waves per (xcc, se, cu, simd)
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 7
    key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
    _, counts = np.unique(key, return_counts=True)
    rec["waves_per_simd_histogram"] = {str(k): int(v) for k, v in zip(*np.unique(counts, return_counts=True))}
    rec["simds_used"] = int(len(counts))
    if counts.max() == 2 and counts.min() == 2:  # one round: the two waves of every SIMD as (shorter, longer)
        order = np.argsort(key, kind="stable")
        pair = dur[order].reshape(-1, 2)
        lo, hi = pair.min(axis=1), pair.max(axis=1)
        slot = (hw & 0xF)[order].reshape(-1, 2)
        rec["pairs"] = {"shorter_mean_us": float(lo.mean()), "shorter_min_us": float(lo.min()), "shorter_max_us": float(lo.max()), "longer_mean_us": float(hi.mean()),
                        "longer_min_us": float(hi.min()), "longer_max_us": float(hi.max()),
                        "slot_pairs": {str(k): int(v) for k, v in zip(*np.unique(slot.min(axis=1) * 16 + slot.max(axis=1), return_counts=True))},
                        "same_parity_pairs": int(((slot[:, 0] ^ slot[:, 1]) & 1 == 0).sum())}
    out = {str(units): rec}  # keep the largest unit count whose stamps are all set = the launch's own
print(json.dumps(out, indent=1))
s.close()
