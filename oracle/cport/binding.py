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

    def close(self):
        if self.h:
            self.lib.cport_setup_destroy(self.h)
            self.h = None


def time_commitment(lib_unused, setup_path, sample_blobs, seed, gpu_outputs=None):
    """bench.py's cpu_baseline leg: blobs/s of the C port on this host, single
    thread and all cores, on a bounded sample of the same synthetic workload."""
    from oracle.pyref import synth

    lib = load()
    cores = os.cpu_count() or 1
    cs = CSetup(lib, setup_path, subgroup_checks=False, threads=1)
    # calibrate on one blob, then size the sample for ~10 s single-threaded + ~10 s threaded
    blob0 = synth.blob_bytes(seed, 0)
    t1, _ = cs.time_commitments(blob0, 1, 1)
    n = sample_blobs or max(2, min(64, int(8.0 / max(t1, 1e-3))))
    blobs = b"".join(synth.blob_bytes(seed, b) for b in range(n))
    t_single, out_single = cs.time_commitments(blobs, n, 1)
    cs.set_threads(cores)
    reps = max(1, int(10.0 / max(t_single / cores * 1.3, 1e-3)))
    reps = min(reps, 50)
    t_multi, out_multi = cs.time_commitments(blobs, n, reps)
    cs.close()
    assert out_single == out_multi, "C port: threaded result differs from single-threaded"
    res = {
        "value": n * reps / t_multi,
        "unit": "blobs/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d synthetic blobs (seed 0x%x, indices 0..%d) x %d passes, all %d host cores; single-thread: %.2f blobs/s"
        % (n, seed, n - 1, reps, cores, n / t_single),
        "single_thread_value": n / t_single,
        "label": "CPU restatement of kateth/blst path (C, 64-bit limbs, Pippenger c=10, per-call base re-normalisation) -- not kateth itself",
        "outputs": out_single.hex() if n <= 8 else None,
        "n": n,
    }
    res["_raw_outputs"] = out_single
    return res
