"""ctypes binding of oracle/cport/libkzg_cport.so (TEST INFRASTRUCTURE: checker
and timed CPU baseline; never imported by kateth_amd)."""
import ctypes
import json
import os
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libkzg_cport.so")


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def load():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "kzg_cport.c")):
        build()
    lib = ctypes.CDLL(SO)
    lib.cport_setup_create.restype = ctypes.c_int
    lib.cport_setup_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    lib.cport_setup_destroy.argtypes = [ctypes.c_void_p]
    lib.cport_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.cport_blob_to_commitment.restype = ctypes.c_int
    lib.cport_blob_to_commitment.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
    lib.cport_time_commitments.restype = ctypes.c_double
    lib.cport_time_commitments.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    lib.cport_time_commitments_blob_parallel.restype = ctypes.c_double
    lib.cport_time_commitments_blob_parallel.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    lib.cport_verify_batch_prepairing.restype = ctypes.c_int
    lib.cport_verify_batch_prepairing.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    lib.cport_time_verify_prepairing.restype = ctypes.c_double
    lib.cport_time_verify_prepairing.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    lib.cport_sha256.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    return lib


def g1_lagrange_bytes(setup_path):
    raw = json.load(open(setup_path))
    return b"".join(bytes.fromhex(s[2:] if s.startswith("0x") else s) for s in raw["g1_lagrange"])


class CSetup:
    def __init__(self, lib, setup_path, subgroup_checks=False, threads=1):
        self.lib = lib
        h = ctypes.c_void_p()
        rc = lib.cport_setup_create(ctypes.byref(h), g1_lagrange_bytes(setup_path), int(subgroup_checks), threads)
        if rc:
            raise RuntimeError("cport_setup_create: point %d rejected with code %d" % (rc // 16 - 1, rc % 16))
        self.h = h

    def set_threads(self, n):
        self.lib.cport_set_threads(self.h, n)

    def blob_to_commitment(self, blob: bytes):
        out = ctypes.create_string_buffer(48)
        st = self.lib.cport_blob_to_commitment(self.h, blob, out)
        return st, out.raw

    def time_commitments(self, blobs: bytes, n: int, reps: int = 1, compress: bool = True):
        out = ctypes.create_string_buffer(48 * n)
        secs = self.lib.cport_time_commitments(self.h, blobs, n, reps, int(compress), out)
        return secs, out.raw

    def time_commitments_blob_parallel(self, blobs: bytes, n: int, reps: int, threads: int):
        out = ctypes.create_string_buffer(48 * n)
        secs = self.lib.cport_time_commitments_blob_parallel(self.h, blobs, n, reps, threads, out)
        return secs, out.raw

    def verify_batch_prepairing(self, blobs: bytes, commitments: bytes, proofs: bytes, n: int, batch_inverse: bool = False):
        """reference-literal verify_blob_proof_batch up to the pairing: (rc, z bytes, y bytes, A48, B48)"""
        z = ctypes.create_string_buffer(32 * n)
        y = ctypes.create_string_buffer(32 * n)
        a = ctypes.create_string_buffer(48)
        b = ctypes.create_string_buffer(48)
        rc = self.lib.cport_verify_batch_prepairing(self.h, blobs, commitments, proofs, n, int(batch_inverse), z, y, a, b)
        return rc, z.raw, y.raw, a.raw, b.raw

    def time_verify_prepairing(self, blobs: bytes, commitments: bytes, proofs: bytes, n: int, batch_inverse: bool = False) -> float:
        return self.lib.cport_time_verify_prepairing(self.h, blobs, commitments, proofs, n, int(batch_inverse))

    def close(self):
        if self.h:
            self.lib.cport_setup_destroy(self.h)
            self.h = None


def host_cores(cap: int = 16) -> int:
    """threads the baseline may use: the process's CPU affinity, capped at the GPU box's
    per-GPU CPU share (16)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, cap))


def time_commitment(lib_unused, setup_path, sample_blobs, seed, gpu_outputs=None):
    """bench.py's cpu_baseline leg: blobs/s of the C port on this host on a bounded
    sample of the same synthetic workload (about 10-30 s of CPU work in total):
      (i)   one thread, blobs one after another                     (benches/kzg.rs:35-37 shape)
      (ii)  reference-like: blobs one after another, MSM tiles spread over `cores` threads
            (what kateth + blst's thread pool does, src/bls.rs:434)
      (iii) blob-parallel: `cores` threads, each a single-threaded MSM on its own blobs
    `value` is the best of (ii) and (iii)."""
    from oracle.pyref import synth

    lib = load()
    cores = host_cores()
    cs = CSetup(lib, setup_path, subgroup_checks=False, threads=1)
    blob0 = synth.blob_bytes(seed, 0)
    t1, _ = cs.time_commitments(blob0, 1, 1)  # calibrate
    n = sample_blobs or max(cores, min(64, int(6.0 / max(t1, 1e-3))))
    blobs = b"".join(synth.blob_bytes(seed, b) for b in range(n))
    t_single, out_single = cs.time_commitments(blobs, n, 1)
    cs.set_threads(cores)
    reps_tiled = max(1, min(20, int(6.0 / max(t_single / min(cores, 8), 1e-3))))
    t_tiled, out_tiled = cs.time_commitments(blobs, n, reps_tiled)
    reps_par = max(1, min(50, int(8.0 * cores / max(t_single, 1e-3))))
    t_par, out_par = cs.time_commitments_blob_parallel(blobs, n, reps_par, cores)
    cs.close()
    assert out_single == out_tiled == out_par, "C port: threaded result differs from single-threaded"
    v_single, v_tiled, v_par = n / t_single, n * reps_tiled / t_tiled, n * reps_par / t_par
    res = {
        "value": max(v_tiled, v_par),
        "unit": "blobs/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d synthetic blobs (seed 0x%x, indices 0..%d): 1 pass single-threaded (%.2f blobs/s), %d passes with the MSM tiled over %d threads "
        "(%.2f blobs/s, reference-like), %d passes blob-parallel on %d threads (%.2f blobs/s)"
        % (n, seed, n - 1, v_single, reps_tiled, cores, v_tiled, reps_par, cores, v_par),
        "single_thread_value": v_single,
        "reference_like_value": v_tiled,
        "blob_parallel_value": v_par,
        "label": "CPU restatement of kateth/blst path (C, 64-bit limbs, Pippenger c=10 signed digits, per-call base re-normalisation) -- not kateth itself",
        "n": n,
    }
    res["_raw_outputs"] = out_single
    return res


def time_verify(setup_path, blobs: bytes, commitments: bytes, proofs: bytes, n: int):
    """cpu_baseline leg for verify_blob_kzg_proof_batch: the reference's algorithm (sequential blobs, one
    Euclidean inversion per element, naive lincombs) up to the single pairing, on the first n triples
    of the benchmark's own inputs; also the labelled batch-inversion variant."""
    lib = load()
    cs = CSetup(lib, setup_path, subgroup_checks=False, threads=1)
    t_ref = cs.time_verify_prepairing(blobs, commitments, proofs, n, False)
    t_batch = cs.time_verify_prepairing(blobs, commitments, proofs, n, True)
    cs.close()
    return {
        "value": n / t_ref,
        "unit": "blobs/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d (blob, commitment, proof) triples of the benchmark batch, single thread like the reference's loop (src/kzg/setup.rs:235-242); "
        "per-element Euclidean inversion: %.2f blobs/s; with batch inversion instead (not what the reference does): %.2f blobs/s; the final pairing (constant per call) is not included"
        % (n, n / t_ref, n / t_batch),
        "batch_inverse_variant_value": n / t_batch,
        "label": "CPU restatement of kateth/blst path (C port) -- not kateth itself",
    }
