"""N > 1 path on CPU: world_size-2 gloo run of the sharding / gather /
first-error orchestration in kateth_amd/dist.py (the GPU engine is replaced by
a deterministic stand-in so that only the multi-rank logic is under test)."""
import hashlib
import os
import struct

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kateth_amd import dist as kdist


def test_shard_range_covers_everything():
    for n in (0, 1, 5, 4096, 65536, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [kdist.shard_range(n, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == n
            pos = 0
            for first, count in spans:
                assert first == min(pos, n) or count == 0
                pos += count


def test_merge_first_error_order():
    ok = [-1, 0, -1, 0, -1, 0]
    assert kdist.merge_first_error([ok, ok], [0, 10]) == (0, -1)
    # rank 1 has a bad blob, rank 0 a bad commitment at a lower index: the blob wins
    assert kdist.merge_first_error([[-1, 0, 2, 3, -1, 0], [5, 2, -1, 0, -1, 0]], [0, 10]) == (2, 15)
    # two bad commitments: lowest global index wins
    assert kdist.merge_first_error([[-1, 0, 9, 4, -1, 0], [-1, 0, 0, 5, -1, 0]], [0, 10]) == (4, 9)
    assert kdist.merge_first_error([[-1, 0, -1, 0, 3, 5], ok], [0, 10]) == (5, 3)


class FakeSetup:
    """stand-in for kateth_amd.Setup: per-item 'commitment' = SHA-256 prefix; verification is
    true iff no item is flagged bad.  Mirrors the method shapes dist.py uses."""

    def __init__(self, items):
        self.items = items  # list of (payload, bad_code)
        self.sess = {}

    def verify_phase1_dev(self, d_blobs, d_c, d_p, n_local, stream=0):
        err6 = [-1, 0, -1, 0, -1, 0]
        for k, (_, bad) in enumerate(self.items):
            if bad and err6[2] < 0:
                err6[2], err6[3] = k, bad
        root = hashlib.sha256(b"".join(p for p, _ in self.items)).digest()
        return object(), root, err6

    def verify_phase2_dev(self, sess, roots, first_index, n_total):
        return hashlib.sha256(roots + struct.pack("<qq", first_index, n_total)).digest() * 6

    def verify_session_destroy(self, sess):
        pass

    def verify_batch_finish(self, partials):
        self.last_partials = partials
        return True


def _worker(rank, world, port, bad_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total = 10
    first, count = kdist.shard_range(n_total, rank, world)
    # commitments gather: rank-ordered concatenation
    local = torch.tensor([first + i for i in range(count) for _ in range(48)], dtype=torch.uint8)
    if count < (n_total + world - 1) // world:  # pad the last shard like bench.py does
        local = torch.cat([local, torch.zeros(((n_total + world - 1) // world - count) * 48, dtype=torch.uint8)])
    gathered = kdist.all_gather_bytes(local, world)
    items = [(bytes([first + i]) * 8, 3 if (rank == bad_rank and i == 1) else 0) for i in range(count)]
    fs = FakeSetup(items)
    try:
        ok = kdist.verify_blob_proof_batch_sharded(fs, 0, 0, 0, count, first, n_total, rank, world, torch.device("cpu"))
        res = ("ok", ok, fs.last_partials)
    except Exception as err:  # noqa: BLE001
        res = ("err", type(err).__name__, str(err))
    q.put((rank, gathered[: n_total * 48].tolist(), res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bad_rank", [-1, 1])
def test_world2_gloo(bad_rank):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + (7 if bad_rank >= 0 else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, bad_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [i for i in range(10) for _ in range(48)]
    for rank, gathered, res in outs:
        assert gathered == want  # global blob order is preserved by the rank-ordered gather
    if bad_rank < 0:
        assert outs[0][2][0] == "ok" and outs[0][2][1] is True
        assert outs[0][2][2] == outs[1][2][2]  # both ranks finished over identical gathered partials
    else:
        for _, _, res in outs:  # every rank raises the same reference-shaped error
            assert res[0] == "err" and res[1] == "KzgError" and "InvalidEncoding" in res[2]


def _bench(*argv):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), capture_output=True, text=True, timeout=120, env=env)


def test_bench_dry_run_prints_every_ranks_memory_plan():
    """`bench.py --dry-run` (no GPU, no torch): BASELINE configs[4] = 2^20 blobs over 8 GPUs is 131,072 blobs = 16 GiB per
    rank beside the 192-GiB table and fits an MI355X (288 GiB as the device reports it); a batch that cannot fit fails BEFORE any allocation"""
    import json

    out = _bench("--dry-run", "--workload", "commit", "--batch", "131072", "--gpus", "8")
    assert out.returncode == 0, out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["blobs_total"] == 1 << 20 and len(rec["ranks"]) == 8
    assert [r["first_blob"] for r in rec["ranks"]] == [131072 * k for k in range(8)]
    r0 = rec["ranks"][0]
    assert r0["fits"] and (r0["table_class"], r0["plane_groups"]) == (22, 8)
    assert r0["blobs"] == 16 << 30 and r0["table"] >= 192 * (1 << 30) and 6 << 30 <= r0["workspace"] < 7 << 30  # three workspace slots of 2.2 GiB
    assert r0["resident_total"] <= r0["hbm_total"] - (2 << 30)
    assert rec["exchange_bytes_per_step_per_rank"] == 131072 * 48
    # a part with 160 GiB of HBM steps down to 4 plane groups; a batch that cannot fit is refused with exit code 3
    small = json.loads(_bench("--dry-run", "--workload", "commit", "--batch", "4096", "--assume-hbm-gib", "160").stdout.strip().splitlines()[-1])
    assert (small["ranks"][0]["table_class"], small["ranks"][0]["plane_groups"]) == (22, 4)
    bad = _bench("--dry-run", "--workload", "commit", "--batch", "800000", "--gpus", "8")
    assert bad.returncode == 3 and "does not fit" in bad.stderr
