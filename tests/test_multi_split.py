"""Group contexts without a GPU: the split of a batch over the members and the merge of their first-error records
(kateth_amd/csrc/multi_split.hpp, compiled into the CPU test build) against the Python restatement of the same rules that the
one-process-per-GPU path uses (kateth_amd/dist.py) and against a direct restatement of the reference's order
(src/kzg/setup.rs:259-271: every blob is parsed before any commitment, every commitment before any proof)."""
import ctypes
import os
import random

import pytest

from kateth_amd import dist


@pytest.fixture(scope="module")
def hm():
    import __graft_entry__ as g

    g.build_hostmath()
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostmath", "libhostmath.so"))
    lib.hm_multi_shares.restype = ctypes.c_int
    lib.hm_multi_shares.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    lib.hm_multi_first_error.restype = ctypes.c_int32
    lib.hm_multi_first_error.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int32)]
    return lib


def shares(hm, n, members, rotate=0):
    buf = (ctypes.c_uint64 * (3 * 64))()
    k = hm.hm_multi_shares(n, members, rotate, buf, 64)
    return [(buf[3 * j], buf[3 * j + 1], buf[3 * j + 2]) for j in range(k)]


def test_shares_cover_the_batch_like_shard_range(hm):
    for members in (1, 2, 3, 7, 8):
        for n in (0, 1, 2, 7, 8, 9, 63, 64, 65, 4096, 4099, 65536, 1 << 20):
            got = shares(hm, n, members)
            if n >= members:
                want = [(k,) + dist.shard_range(n, k, members) for k in range(members)]
                assert got == [w for w in want if w[2] > 0], (n, members)
            # contiguous, disjoint, complete, in global order
            assert [s[1] for s in got] == [sum(t[2] for t in got[:j]) for j in range(len(got))]
            assert sum(s[2] for s in got) == n
            assert all(s[2] > 0 and s[0] < members for s in got)


def test_small_calls_rotate_over_the_members(hm):
    for rotate in range(9):
        got = shares(hm, 3, 8, rotate)
        assert [s[0] for s in got] == [(rotate + i) % 8 for i in range(3)] and [(s[1], s[2]) for s in got] == [(0, 1), (1, 1), (2, 1)]
    assert shares(hm, 1, 4, 7) == [(3, 0, 1)]


def test_first_error_merge_matches_the_reference_order(hm):
    rng = random.Random(0x5EED)
    for trial in range(400):
        members = rng.choice([1, 2, 3, 4, 8])
        n = rng.choice([members, members + 1, 37, 100, 4099])
        sh = shares(hm, n, members)
        # random statuses for blobs / commitments / proofs (mostly clean)
        stat = [[0] * n for _ in range(3)]
        for kind in range(3):
            for _ in range(rng.choice([0, 0, 1, 2, 3])):
                stat[kind][rng.randrange(n)] = rng.choice([2] if kind == 0 else [3, 4, 5])
        # what every member's phase 1 reports: local index and code of its first error of each kind
        err6 = []
        for _, first, count in sh:
            rec = []
            for kind in range(3):
                local = next((i for i in range(count) if stat[kind][first + i]), -1)
                rec += [local, stat[kind][first + local] if local >= 0 else 0]
            err6 += rec
        arr = (ctypes.c_int32 * len(err6))(*err6)
        got = hm.hm_multi_first_error(n, members, 0, arr)
        want = 0
        for kind in range(3):  # the reference: collect blobs, then commitments, then proofs; the first Err wins
            bad = next((c for c in stat[kind] if c), 0)
            if bad:
                want = bad
                break
        assert got == want, (trial, members, n)
        recs = [err6[6 * j:6 * j + 6] for j in range(len(sh))]
        assert dist.merge_first_error(recs, [s[1] for s in sh])[0] == want
