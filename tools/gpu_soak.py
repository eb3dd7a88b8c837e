#!/usr/bin/env python3
"""Long on-device cross-check of the inline-asm multiply chains against the compiler-scheduled multiply, and of the
radix-2^28 / 2^29 multipliers (product, squaring, two products with one reduction) against the 32-bit-limb multiplier."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kateth_amd  # noqa: E402

s = kateth_amd.Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"), window_bits=6)
total = 0
bad = 0
t0 = time.time()
for lanes, iters in ((64, 400000), (256 * 4 * 64, 8000), (256 * 4 * 64 * 2, 8000), (256 * 4 * 64 * 4, 4000)):
    b = s.selftest_field_mul(lanes, iters)
    total += 8 * lanes * iters  # per iteration: Fp and Fr asm-vs-C, plus 3 radix-2^28 and 3 radix-2^29 cross-checks
    bad += b
    print("lanes=%d iters=%d mismatches=%d  (%.1f s)" % (lanes, iters, b, time.time() - t0), flush=True)
print("TOTAL cross-checks (Fp, Fr, fp28 x3, fr29 x3): %.3e  mismatches: %d" % (total, bad))
s.close()
sys.exit(1 if bad else 0)
