"""Host-side mirror of kateth's `kzg::Setup` over the C ABI (ctypes).

Method names, argument meaning and error behaviour follow the reference:

    Setup.load_json(path)                        src/kzg/setup.rs:46-82
    Setup.blob_to_commitment(blob)               src/kzg/setup.rs:167-171   (+ compress, src/bls.rs:491-503)
    Setup.blob_proof(blob, commitment48)         src/kzg/setup.rs:177-183
    Setup.proof(blob, z32)                       src/kzg/setup.rs:185-194
    Setup.verify_proof(proof, commitment, z, y)  src/kzg/setup.rs:96-113
    Setup.verify_blob_proof(blob, c, p)          src/kzg/setup.rs:208-221
    Setup.verify_blob_proof_batch(blobs, cs, ps) src/kzg/setup.rs:247-275

Points cross this boundary in their 48-byte compressed form (what every caller
of the reference does next: benches/kzg.rs:25-32, src/kzg/setup.rs:341-343).
The `*_batch` / `*_dev` methods expose the batch-level C entry points directly;
the single-item methods are batches of one.
"""
from __future__ import annotations

import ctypes
import json
import os
from typing import List, Optional, Sequence, Tuple

BYTES_PER_BLOB = 131072
_HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------------------
# error types (src/blob.rs:6-10, src/bls.rs:21-50, src/kzg/mod.rs:15-31)
# ---------------------------------------------------------------------------
class BlobError(Exception):
    def __init__(self, kind: str):
        super().__init__("blob::Error::" + kind)
        self.kind = kind


class FiniteFieldError(Exception):
    def __init__(self, kind: str):
        super().__init__("bls::FiniteFieldError::" + kind)
        self.kind = kind


class ECGroupError(Exception):
    def __init__(self, kind: str):
        super().__init__("bls::ECGroupError::" + kind)
        self.kind = kind


class BlsError(Exception):
    """`bls::Error` -- wraps FiniteField / ECGroup."""

    def __init__(self, inner: Exception):
        super().__init__(str(inner))
        self.inner = inner


class KzgError(Exception):
    """`kzg::Error` -- `Blob(blob::Error)` or `Bls(bls::Error)`."""

    def __init__(self, inner: Exception):
        super().__init__(str(inner))
        self.inner = inner


class LoadSetupError(Exception):
    pass


class EngineError(RuntimeError):
    """negative return from the C ABI: HIP / argument / device failure."""


_STATUS = {
    1: lambda: BlobError("InvalidLen"),
    2: lambda: BlobError("InvalidFieldElement"),
    3: lambda: ECGroupError("InvalidEncoding"),
    4: lambda: ECGroupError("NotOnCurve"),
    5: lambda: ECGroupError("NotInGroup"),
    6: lambda: FiniteFieldError("InvalidEncoding"),
    7: lambda: FiniteFieldError("NotInFiniteField"),
}


def error_from_status(code: int) -> Exception:
    return _STATUS[code]()


def _kzg_error(code: int) -> KzgError:
    inner = error_from_status(code)
    return KzgError(inner if isinstance(inner, BlobError) else BlsError(inner))


# ---------------------------------------------------------------------------
# library loading -- fails loudly, no fallback
# ---------------------------------------------------------------------------
class _Config(ctypes.Structure):
    """kzg_config (include/kateth_amd.h)"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("window_bits", ctypes.c_int32), ("flags", ctypes.c_int32), ("plane_groups", ctypes.c_int32),
                ("table_budget_bytes", ctypes.c_uint64), ("devices", ctypes.POINTER(ctypes.c_int32)), ("ndev", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]

    @classmethod
    def new(cls, device=0, window_bits=0, flags=0, plane_groups=0, table_budget_bytes=0, devices=None, ndev=0, reserved=0):
        """KZG_CONFIG_INIT + fields: struct_size = sizeof(kzg_config), which the library checks"""
        return cls(ctypes.sizeof(cls), device, window_bits, flags, plane_groups, table_budget_bytes, devices, ndev, reserved)


CFG_TABLE_MAX = 0x1    # KZG_CFG_TABLE_MAX: the automatic table choice may take the largest table the device has room for (192 GiB)
CFG_BUILD_ASYNC = 0x2  # KZG_CFG_BUILD_ASYNC: usable on a small first-use table at once, the chosen table is built in the background
ALL_DEVICES = 0xFFFFFFFF  # KZG_ALL_DEVICES


def library_path() -> str:
    return os.environ.get("KATETH_AMD_LIB", os.path.join(_HERE, "libkateth_amd.so"))


_LIB = None
_ALT_LIBS = {}

_u8p = ctypes.c_void_p
_i32p = ctypes.POINTER(ctypes.c_int32)
_vpp = ctypes.POINTER(ctypes.c_void_p)
_u64p = ctypes.POINTER(ctypes.c_uint64)

_SIGNATURES = {
    "kzg_last_error": (ctypes.c_char_p, []),
    "kzg_last_error_code": (ctypes.c_int32, []),
    "kzg_ctx_create": (ctypes.c_int32, [_u8p, _u8p, ctypes.POINTER(_Config), ctypes.POINTER(ctypes.c_void_p)]),
    "kzg_ctx_create_multi": (ctypes.c_int32, [_u8p, _u8p, ctypes.POINTER(ctypes.c_int32), ctypes.c_uint32, ctypes.POINTER(_Config), ctypes.POINTER(ctypes.c_void_p)]),
    "kzg_ctx_destroy": (None, [ctypes.c_void_p]),
    "kzg_device_count": (ctypes.c_int32, []),
    "kzg_ctx_members": (ctypes.c_uint32, [ctypes.c_void_p]),
    "kzg_ctx_member_device": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint32]),
    "kzg_ctx_member": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_uint32]),
    "kzg_ctx_ready": (ctypes.c_int32, [ctypes.c_void_p]),
    "kzg_ctx_wait_ready": (ctypes.c_int32, [ctypes.c_void_p]),
    "kzg_ctx_window_bits": (ctypes.c_int32, [ctypes.c_void_p]),
    "kzg_ctx_msm_kernel_name": (ctypes.c_char_p, [ctypes.c_void_p]),
    "kzg_ctx_plane_groups": (ctypes.c_int32, [ctypes.c_void_p]),
    "kzg_ctx_table_bytes": (ctypes.c_uint64, [ctypes.c_void_p]),
    "kzg_blob_to_commitment_batch": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint64, _u8p, _i32p]),
    "kzg_blob_to_commitment_batch_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kzg_blob_to_commitment_batch_affine": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint64, _u8p, _i32p]),
    "kzg_compute_blob_proof_batch_affine": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint64, _u8p, _i32p]),
    "kzg_compute_proof_batch_affine": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint64, _u8p, _u8p, _i32p]),
    "kzg_compute_blob_proof_batch": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint64, _u8p, _i32p]),
    "kzg_compute_blob_proof_batch_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kzg_compute_proof_batch": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint64, _u8p, _u8p, _i32p]),
    "kzg_verify_blob_proof_batch": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p, _u8p, ctypes.c_uint64, _i32p]),
    "kzg_verify_blob_proof_batch_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, _i32p, ctypes.c_void_p]),
    "kzg_verify_blob_proof": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p, _u8p, _i32p]),
    "kzg_verify_proof": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p, _u8p, _u8p, _i32p]),
    "kzg_g1_decompress_batch": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint64, _u8p, _i32p]),
    "kzg_evaluate_blobs": (ctypes.c_int32, [ctypes.c_void_p, _u8p, _u8p, ctypes.c_uint64, _u8p, _i32p]),
    "kzg_verify_phase1_dev": (
        ctypes.c_int32,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, _u8p, _i32p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p],
    ),
    "kzg_blob_to_commitment_batch_group_dev": (ctypes.c_int32, [ctypes.c_void_p, _vpp, _u64p, _vpp, _vpp, _vpp]),
    "kzg_compute_blob_proof_batch_group_dev": (ctypes.c_int32, [ctypes.c_void_p, _vpp, _vpp, _u64p, _vpp, _vpp, _vpp]),
    "kzg_verify_blob_proof_batch_group_dev": (ctypes.c_int32, [ctypes.c_void_p, _vpp, _vpp, _vpp, _u64p, _i32p, _vpp]),
    "kzg_recommended_env": (ctypes.c_char_p, []),
    "kzg_ctx_workspace_bytes": (ctypes.c_uint64, [ctypes.c_void_p, ctypes.c_uint32]),
    "kzg_verify_phase2_dev": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, _u8p]),
    "kzg_verify_session_destroy": (None, [ctypes.c_void_p]),
    "kzg_verify_session_zy": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, _u8p, _u8p]),
    "kzg_verify_batch_finish": (ctypes.c_int32, [ctypes.c_void_p, _u8p, ctypes.c_uint64, _i32p]),
    "kzg_synth_blobs_dev": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]),
    "kzg_profile_begin": (ctypes.c_int32, [ctypes.c_void_p]),
    "kzg_profile_end": (ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]),
    "kzg_profile_end_kinds": (ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]),
    "kzg_profile_kind_name": (ctypes.c_char_p, [ctypes.c_int32]),
    "kzg_ctx_adds_per_blob": (ctypes.c_uint64, [ctypes.c_void_p]),
    "kzg_selftest_field_mul": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
    "kzg_selftest_exception_guard": (ctypes.c_int32, [ctypes.c_int32]),
    "kzg_microbench_fp_mul": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_float)]),
    "kzg_clock_probe_launch": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint32]),
    "kzg_clock_probe_read": (ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "kzg_microbench_valu_issue": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _ensure_hip_runtime():
    """libkateth_amd.so carries no DT_NEEDED for the HIP runtime: it binds to the
    libamdhip64 already in the process.  PyTorch-ROCm wheels bundle their own copy
    and a second runtime in the same process breaks both, so when torch is
    installed its runtime is the one that gets loaded (and promoted to the global
    symbol scope); otherwise the system ROCm runtime is."""
    mode = getattr(ctypes, "RTLD_GLOBAL", 0)
    try:
        import torch  # noqa: F401  (plumbing only: loads torch's HIP runtime)

        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=mode)
            return
    except ImportError:
        pass
    for cand in (os.environ.get("KATETH_AMD_HIP_RUNTIME", ""), "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"):
        if cand:
            try:
                ctypes.CDLL(cand, mode=mode)
                return
            except OSError:
                continue
    raise ImportError("kateth_amd: no HIP runtime (libamdhip64.so) could be loaded")


def load_library(path: Optional[str] = None):
    """dlopen the HIP engine.  Raises if it has not been built -- by design there
    is nothing to fall back to.  `path`: another build of the same C ABI (the tests'
    cross-check build under tests/radix32); the default is the product library."""
    global _LIB
    if path is None and _LIB is not None:
        return _LIB
    if path is not None and path in _ALT_LIBS:
        return _ALT_LIBS[path]
    alt = path is not None
    path = path or library_path()
    if not os.path.exists(path):
        raise ImportError(
            "kateth_amd: HIP engine %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path
        )
    _ensure_hip_runtime()
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export what include/kateth_amd.h declares
        fn.restype = restype
        fn.argtypes = argtypes
    if alt:
        _ALT_LIBS[path] = lib
    else:
        _LIB = lib
    return lib


def _buf(data) -> bytes:
    if isinstance(data, (bytes, bytearray, memoryview)):
        return bytes(data)
    if hasattr(data, "to_bytes") and not isinstance(data, int):  # a Blob
        return data.to_bytes()
    return bytes(bytearray(data))


def _has_noncanonical_element(blob: bytes) -> bool:
    """Blob::from_slice's per-element check (src/blob.rs:32-34 -> src/bls.rs:110-120): some 32-byte big-endian element >= r.
    Host side, error path only: the mirror needs it to name the FIRST failing blob when a later one has the wrong length."""
    r = _R.to_bytes(32, "big")
    return any(blob[k:k + 32] >= r for k in range(0, len(blob) - len(blob) % 32, 32))


def _unhex(s: str) -> bytes:
    """`Bytes` deserialiser (src/bytes.rs:30-37): optional 0x prefix."""
    return bytes.fromhex(s[2:] if s.startswith("0x") else s)


_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # Fr modulus
_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB  # Fp modulus


class Blob:
    """`Blob<4096>` (src/blob.rs:18-76): a validated 131,072-byte blob.  Host-side container only -- every computation
    on it happens in the engine; `from_slice` repeats the reference's parse checks so that the type has its meaning."""

    BYTES = BYTES_PER_BLOB  # Blob::<4096>::BYTES, src/blob.rs:24

    def __init__(self, data: bytes):
        self._data = data

    @classmethod
    def from_slice(cls, data) -> "Blob":
        """src/blob.rs:26-37: InvalidLen unless 131,072 bytes; InvalidFieldElement unless every 32-byte big-endian chunk is < r."""
        data = _buf(data)
        if len(data) != cls.BYTES:
            raise BlobError("InvalidLen")
        rb = _R.to_bytes(32, "big")
        for i in range(0, cls.BYTES, 32):
            if data[i:i + 32] >= rb:  # big-endian: bytewise order is numeric order
                raise BlobError("InvalidFieldElement")
        return cls(data)

    def to_bytes(self) -> bytes:
        """src/blob.rs:39-46"""
        return self._data

    @classmethod
    def random(cls, gen) -> "Blob":
        """src/blob.rs:66-76: each element = SHA-256(512 bytes from `gen`) reduced mod r (`Fr::hash_to`, src/bls.rs:189-205).
        `gen` is anything with randbytes(n) (random.Random) or a callable n -> bytes."""
        import hashlib

        fill = gen.randbytes if hasattr(gen, "randbytes") else gen
        out = bytearray()
        for _ in range(cls.BYTES // 32):
            out += (int.from_bytes(hashlib.sha256(fill(512)).digest(), "big") % _R).to_bytes(32, "big")
        return cls(bytes(out))

    def __bytes__(self):
        return self._data

    def __len__(self):
        return len(self._data)


class P1:
    """`bls::P1` as the reference's producers return it (`Commitment = Proof = P1`, src/kzg/mod.rs:9-10): the engine hands
    the point over as a 96-byte blst_p1_affine image (x || y, little-endian limbs of the 2^384-Montgomery residue; all
    zero = infinity), which is what a Rust caller feeds to blst_p1_from_affine.  `compress` is the caller-side
    `Compress::compress` (src/bls.rs:491-503, a blst CPU call in the reference): a change of encoding of a point the GPU
    already normalised, kept here so the mirror's call sites read like benches/kzg.rs:24-32."""

    COMPRESSED = 48

    def __init__(self, affine96: bytes):
        assert len(affine96) == 96
        self.affine = bytes(affine96)

    def is_inf(self) -> bool:
        return not any(self.affine)

    def compress(self) -> bytes:
        if self.is_inf():
            return bytes([0xC0]) + bytes(47)
        rinv = pow(1 << 384, -1, _P)
        x = int.from_bytes(self.affine[:48], "little") * rinv % _P
        y = int.from_bytes(self.affine[48:], "little") * rinv % _P
        out = bytearray(x.to_bytes(48, "big"))
        out[0] |= 0x80 | (0x20 if y > (_P - 1) // 2 else 0)
        return bytes(out)

    def __eq__(self, other):
        return isinstance(other, P1) and self.affine == other.affine

    def __hash__(self):
        return hash(self.affine)


class Setup:
    """`Setup<4096, 65>` (src/kzg/setup.rs:37-42) resident on one MI355X."""

    G1 = 4096
    G2 = 65

    def __init__(self, handle: int, lib):
        self._h = ctypes.c_void_p(handle)
        self._lib = lib

    # -- construction --------------------------------------------------------
    @classmethod
    def load_json(cls, path, device: int = 0, window_bits: int = 0, lib_path: Optional[str] = None, plane_groups: int = 0, devices=None,
                  table_max: bool = False, build_async: bool = False, table_budget_bytes: int = 0) -> "Setup":
        """`Setup::load_json` (src/kzg/setup.rs:46-82).  window_bits = 0: the engine picks the fastest table class within the
        budget (default 100 GiB -> the 96-GiB table; `table_max` lifts the cap -> 192 GiB on an idle MI355X) that the device
        has room for (include/kateth_amd.h, kzg_config).  `devices`: a list of HIP ordinals, or "all" -- a GROUP context whose
        host-buffer methods shard every batch over the listed GPUs.  `build_async`: return as soon as a small first-use table
        stands; the chosen table is built in the background (`ready`, `wait_ready`)."""
        try:
            with open(path) as fh:
                raw = json.load(fh)
        except OSError as err:
            raise LoadSetupError("Io(%s)" % err)
        except ValueError as err:
            raise LoadSetupError("Serde(%s)" % err)
        try:
            g1 = [_unhex(s) for s in raw["g1_lagrange"]]
            g2 = [_unhex(s) for s in raw["g2_monomial"]]
        except (KeyError, ValueError, AttributeError) as err:
            raise LoadSetupError("Serde(%s)" % err)
        return cls.from_bytes(g1, g2, device=device, window_bits=window_bits, lib_path=lib_path, plane_groups=plane_groups, devices=devices,
                              table_max=table_max, build_async=build_async, table_budget_bytes=table_budget_bytes)

    @classmethod
    def from_bytes(cls, g1_lagrange: Sequence[bytes], g2_monomial: Sequence[bytes], device: int = 0, window_bits: int = 0,
                   lib_path: Optional[str] = None, plane_groups: int = 0, devices=None, table_max: bool = False, build_async: bool = False,
                   table_budget_bytes: int = 0) -> "Setup":
        if len(g1_lagrange) != cls.G1:
            raise LoadSetupError("InvalidLenG1Lagrange")  # src/kzg/setup.rs:52-54
        if len(g2_monomial) != cls.G2:
            raise LoadSetupError("InvalidLenG2Monomial")  # src/kzg/setup.rs:55-57
        if any(len(p) != 48 for p in g1_lagrange) or any(len(p) != 96 for p in g2_monomial):
            raise LoadSetupError("Bls(ECGroup(InvalidEncoding))")
        lib = load_library(lib_path)
        flags = (CFG_TABLE_MAX if table_max else 0) | (CFG_BUILD_ASYNC if build_async else 0)
        cfg = _Config.new(device, window_bits, flags, plane_groups, table_budget_bytes, None, 0, 0)
        keep = None
        if devices is not None:
            if isinstance(devices, str):
                assert devices == "all", devices
                cfg.ndev = ALL_DEVICES
            else:
                keep = (ctypes.c_int32 * len(devices))(*devices)
                cfg.devices = ctypes.cast(keep, ctypes.POINTER(ctypes.c_int32))
                cfg.ndev = len(devices)
        out = ctypes.c_void_p()
        rc = lib.kzg_ctx_create(b"".join(g1_lagrange), b"".join(g2_monomial), ctypes.byref(cfg), ctypes.byref(out))
        if rc in (-4, -5):  # LoadSetupError::Bls(bls::Error::ECGroup(..)), src/kzg/setup.rs:59-72
            code = lib.kzg_last_error_code()
            kind = str(error_from_status(code)) if code in _STATUS else "ECGroup"
            raise LoadSetupError("Bls(%s): %s" % (kind, lib.kzg_last_error().decode()))
        if rc == -6:  # no counterpart in the reference: a setup whose points cancel within a comb block (include/kateth_amd.h)
            raise LoadSetupError("Unsupported: %s" % lib.kzg_last_error().decode())
        if rc != 0:
            raise EngineError("kzg_ctx_create failed (%d): %s" % (rc, lib.kzg_last_error().decode()))
        return cls(out.value, lib)

    # -- group contexts / background build ------------------------------------
    @property
    def members(self) -> int:
        """devices this context shards host-buffer batches over (1 = a single-device context)"""
        return self._lib.kzg_ctx_members(self._h)

    def member_device(self, k: int) -> int:
        return self._lib.kzg_ctx_member_device(self._h, k)

    def member(self, k: int) -> "Setup":
        """member k as a single-device Setup (borrowed: it lives as long as this context; close() on it is a no-op)"""
        h = self._lib.kzg_ctx_member(self._h, k)
        if not h:
            raise IndexError(k)
        m = Setup(h, self._lib)
        m._borrowed = True
        m._owner = self
        return m

    @property
    def ready(self) -> bool:
        return bool(self._lib.kzg_ctx_ready(self._h))

    def wait_ready(self):
        self._check(self._lib.kzg_ctx_wait_ready(self._h), "kzg_ctx_wait_ready")

    def close(self):
        if self._h:
            if not getattr(self, "_borrowed", False):
                self._lib.kzg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self) -> int:
        return self._h.value

    @property
    def window_bits(self) -> int:
        return self._lib.kzg_ctx_window_bits(self._h)

    @property
    def plane_groups(self) -> int:
        return self._lib.kzg_ctx_plane_groups(self._h)

    @property
    def msm_kernel_name(self) -> str:
        return self._lib.kzg_ctx_msm_kernel_name(self._h).decode()

    @property
    def table_bytes(self) -> int:
        return self._lib.kzg_ctx_table_bytes(self._h)

    def workspace_bytes(self):
        """bytes of the three commitment / proof workspace slots as allocated so far"""
        return [int(self._lib.kzg_ctx_workspace_bytes(self._h, k)) for k in range(3)]

    def _check(self, rc: int, what: str):
        if rc < 0:
            raise EngineError("%s failed (%d): %s" % (what, rc, self._lib.kzg_last_error().decode()))

    # -- batch entry points (host buffers) --------------------------------------
    def blob_to_commitment_batch(self, blobs: bytes, n: Optional[int] = None):
        """n concatenated blobs -> (n*48 bytes, [status])."""
        blobs = _buf(blobs)
        n = len(blobs) // BYTES_PER_BLOB if n is None else n
        if len(blobs) != n * BYTES_PER_BLOB:
            raise BlobError("InvalidLen")
        out = ctypes.create_string_buffer(48 * n)
        status = (ctypes.c_int32 * n)()
        rc = self._lib.kzg_blob_to_commitment_batch(self._h, blobs, n, ctypes.cast(out, ctypes.c_void_p), status)
        self._check(rc, "kzg_blob_to_commitment_batch")
        return out.raw, list(status)

    def compute_blob_proof_batch(self, blobs: bytes, commitments: bytes):
        blobs, commitments = _buf(blobs), _buf(commitments)
        n = len(commitments) // 48
        if len(blobs) != n * BYTES_PER_BLOB or len(commitments) != 48 * n:
            raise BlobError("InvalidLen")
        out = ctypes.create_string_buffer(48 * n)
        status = (ctypes.c_int32 * n)()
        rc = self._lib.kzg_compute_blob_proof_batch(self._h, blobs, commitments, n, ctypes.cast(out, ctypes.c_void_p), status)
        self._check(rc, "kzg_compute_blob_proof_batch")
        return out.raw, list(status)

    def compute_proof_batch(self, blobs: bytes, zs: bytes):
        blobs, zs = _buf(blobs), _buf(zs)
        n = len(zs) // 32
        if len(blobs) != n * BYTES_PER_BLOB:
            raise BlobError("InvalidLen")
        proofs = ctypes.create_string_buffer(48 * n)
        ys = ctypes.create_string_buffer(32 * n)
        status = (ctypes.c_int32 * n)()
        rc = self._lib.kzg_compute_proof_batch(self._h, blobs, zs, n, ctypes.cast(proofs, ctypes.c_void_p), ctypes.cast(ys, ctypes.c_void_p), status)
        self._check(rc, "kzg_compute_proof_batch")
        return proofs.raw, ys.raw, list(status)

    # -- the same producers returning POINTS (`Commitment = Proof = P1`, src/kzg/mod.rs:9-10) as 96-byte blst_p1_affine images
    def blob_to_commitment_batch_affine(self, blobs: bytes, n: Optional[int] = None):
        blobs = _buf(blobs)
        n = len(blobs) // BYTES_PER_BLOB if n is None else n
        if len(blobs) != n * BYTES_PER_BLOB:
            raise BlobError("InvalidLen")
        out = ctypes.create_string_buffer(96 * n)
        status = (ctypes.c_int32 * n)()
        self._check(self._lib.kzg_blob_to_commitment_batch_affine(self._h, blobs, n, ctypes.cast(out, ctypes.c_void_p), status), "kzg_blob_to_commitment_batch_affine")
        return out.raw, list(status)

    def compute_blob_proof_batch_affine(self, blobs: bytes, commitments: bytes):
        blobs, commitments = _buf(blobs), _buf(commitments)
        n = len(commitments) // 48
        if len(blobs) != n * BYTES_PER_BLOB or len(commitments) != 48 * n:
            raise BlobError("InvalidLen")
        out = ctypes.create_string_buffer(96 * n)
        status = (ctypes.c_int32 * n)()
        self._check(self._lib.kzg_compute_blob_proof_batch_affine(self._h, blobs, commitments, n, ctypes.cast(out, ctypes.c_void_p), status), "kzg_compute_blob_proof_batch_affine")
        return out.raw, list(status)

    def compute_proof_batch_affine(self, blobs: bytes, zs: bytes):
        blobs, zs = _buf(blobs), _buf(zs)
        n = len(zs) // 32
        if len(blobs) != n * BYTES_PER_BLOB:
            raise BlobError("InvalidLen")
        proofs = ctypes.create_string_buffer(96 * n)
        ys = ctypes.create_string_buffer(32 * n)
        status = (ctypes.c_int32 * n)()
        rc = self._lib.kzg_compute_proof_batch_affine(self._h, blobs, zs, n, ctypes.cast(proofs, ctypes.c_void_p), ctypes.cast(ys, ctypes.c_void_p), status)
        self._check(rc, "kzg_compute_proof_batch_affine")
        return proofs.raw, ys.raw, list(status)

    # -- reference-shaped API ----------------------------------------------------
    # `Setup::blob_to_commitment / blob_proof / proof` with the reference's return type (a point), for call sites shaped
    # like benches/kzg.rs:24-32 (`kzg.blob_to_commitment(blob).unwrap().compress(&mut bytes)`)
    def blob_to_commitment_point(self, blob) -> P1:
        blob = _buf(blob)
        if len(blob) != BYTES_PER_BLOB:
            raise BlobError("InvalidLen")
        out, status = self.blob_to_commitment_batch_affine(blob, 1)
        if status[0]:
            raise error_from_status(status[0])
        return P1(out)

    def blob_proof_point(self, blob, commitment: bytes) -> P1:
        blob, commitment = _buf(blob), _buf(commitment)
        if len(blob) != BYTES_PER_BLOB:
            raise KzgError(BlobError("InvalidLen"))
        if len(commitment) != 48:
            raise KzgError(BlsError(ECGroupError("InvalidEncoding")))
        out, status = self.compute_blob_proof_batch_affine(blob, commitment)
        if status[0]:
            raise _kzg_error(status[0])
        return P1(out)

    def proof_point(self, blob, point: bytes):
        """(P1 proof, y32)"""
        blob, point = _buf(blob), _buf(point)
        if len(blob) != BYTES_PER_BLOB:
            raise KzgError(BlobError("InvalidLen"))
        if len(point) != 32:
            raise KzgError(BlsError(FiniteFieldError("InvalidEncoding")))
        proofs, ys, status = self.compute_proof_batch_affine(blob, point)
        if status[0]:
            raise _kzg_error(status[0])
        return P1(proofs), ys

    def blob_to_commitment(self, blob: bytes) -> bytes:
        """`Setup::blob_to_commitment` + `compress`: 48-byte commitment or BlobError."""
        blob = _buf(blob)
        if len(blob) != BYTES_PER_BLOB:
            raise BlobError("InvalidLen")  # src/blob.rs:27-29
        out, status = self.blob_to_commitment_batch(blob, 1)
        if status[0]:
            raise error_from_status(status[0])
        return out

    def blob_proof(self, blob: bytes, commitment: bytes) -> bytes:
        """`Setup::blob_proof` + `compress` (kzg::Error on bad input)."""
        blob, commitment = _buf(blob), _buf(commitment)
        if len(blob) != BYTES_PER_BLOB:
            raise KzgError(BlobError("InvalidLen"))
        if len(commitment) != 48:
            raise KzgError(BlsError(ECGroupError("InvalidEncoding")))
        out, status = self.compute_blob_proof_batch(blob, commitment)
        if status[0]:
            raise _kzg_error(status[0])
        return out

    def proof(self, blob: bytes, point: bytes):
        """`Setup::proof`: (proof48, y32)."""
        blob, point = _buf(blob), _buf(point)
        if len(blob) != BYTES_PER_BLOB:
            raise KzgError(BlobError("InvalidLen"))
        if len(point) != 32:
            raise KzgError(BlsError(FiniteFieldError("InvalidEncoding")))  # src/bls.rs:131-133
        proofs, ys, status = self.compute_proof_batch(blob, point)
        if status[0]:
            raise _kzg_error(status[0])
        return proofs, ys

    def evaluate_blobs(self, blobs: bytes, points32: bytes) -> Tuple[bytes, List[int]]:
        """`Polynomial::evaluate` (src/kzg/poly.rs:10-33) for n (blob, z) pairs through the verification path's evaluation
        kernel: n * 32 big-endian bytes of evaluations and the per-item status (0, BlobError InvalidFieldElement, or
        FiniteFieldError NotInFiniteField for z)."""
        blobs, points32 = _buf(blobs), _buf(points32)
        if len(points32) % 32 or len(blobs) != (len(points32) // 32) * BYTES_PER_BLOB:
            raise KzgError(BlobError("InvalidLen"))
        n = len(points32) // 32
        out = (ctypes.c_uint8 * (32 * max(n, 1)))()
        status = (ctypes.c_int32 * max(n, 1))()
        rc = self._lib.kzg_evaluate_blobs(self._h, blobs, points32, n, out, status)
        self._check(rc, "kzg_evaluate_blobs")
        return bytes(out)[:32 * n], list(status)[:n]

    def decompress_g1_batch(self, points48) -> Tuple[List["P1"], List[int]]:
        """`P1::decompress` (the crate's `Decompress` trait on `Commitment` / `Proof`, src/bls.rs:505-531) for a list (or a
        concatenation) of 48-byte encodings: the points and the per-point status (0 or an ECGroupError code)."""
        data = b"".join(points48) if isinstance(points48, (list, tuple)) else _buf(points48)
        if len(data) % 48:
            raise KzgError(BlsError(ECGroupError("InvalidEncoding")))
        n = len(data) // 48
        out = (ctypes.c_uint8 * (96 * max(n, 1)))()
        status = (ctypes.c_int32 * max(n, 1))()
        rc = self._lib.kzg_g1_decompress_batch(self._h, data, n, out, status)
        self._check(rc, "kzg_g1_decompress_batch")
        raw = bytes(out)
        return [P1(raw[96 * i:96 * i + 96]) for i in range(n)], list(status)[:n]

    def decompress_g1(self, point48: bytes) -> "P1":
        """one point; raises the reference's error (`bls::Error::ECGroup(..)`) for a rejected encoding"""
        if len(point48) != 48:
            raise BlsError(ECGroupError("InvalidEncoding"))
        pts, st = self.decompress_g1_batch(point48)
        if st[0]:
            raise BlsError(error_from_status(st[0]))
        return pts[0]

    def verify_proof(self, proof: bytes, commitment: bytes, point: bytes, evaluation: bytes) -> bool:
        proof, commitment, point, evaluation = _buf(proof), _buf(commitment), _buf(point), _buf(evaluation)
        if len(proof) != 48 or len(commitment) != 48:
            raise KzgError(BlsError(ECGroupError("InvalidEncoding")))
        if len(point) != 32 or len(evaluation) != 32:
            raise KzgError(BlsError(FiniteFieldError("InvalidEncoding")))
        ok = ctypes.c_int32(0)
        rc = self._lib.kzg_verify_proof(self._h, proof, commitment, point, evaluation, ctypes.byref(ok))
        self._check(rc, "kzg_verify_proof")
        if rc > 0:
            raise _kzg_error(rc)
        return bool(ok.value)

    def verify_blob_proof(self, blob: bytes, commitment: bytes, proof: bytes) -> bool:
        blob, commitment, proof = _buf(blob), _buf(commitment), _buf(proof)
        if len(blob) != BYTES_PER_BLOB:
            raise KzgError(BlobError("InvalidLen"))
        if len(commitment) != 48 or len(proof) != 48:
            raise KzgError(BlsError(ECGroupError("InvalidEncoding")))
        ok = ctypes.c_int32(0)
        rc = self._lib.kzg_verify_blob_proof(self._h, blob, commitment, proof, ctypes.byref(ok))
        self._check(rc, "kzg_verify_blob_proof")
        if rc > 0:
            raise _kzg_error(rc)
        return bool(ok.value)

    def verify_blob_proof_batch(self, blobs: Sequence[bytes], commitments: Sequence[bytes], proofs: Sequence[bytes]) -> bool:
        """`Setup::verify_blob_proof_batch`.  Length mismatch panics in the
        reference (src/kzg/setup.rs:256-257) -> AssertionError here."""
        assert len(blobs) == len(commitments), "assertion `left == right` failed"
        assert len(commitments) == len(proofs), "assertion `left == right` failed"
        n = len(blobs)
        # first-error-wins order of the reference: blobs (in index order: `collect` stops at the FIRST blob that fails, whatever its
        # error -- src/kzg/setup.rs:259-262), then commitments, then proofs.  A short blob never reaches the engine (the ABI takes n
        # blobs of 131,072 bytes), so the blobs BEFORE it are checked here the way Blob::from_slice checks them (src/blob.rs:26-37):
        # (blob 0 non-canonical, blob 1 short) is InvalidFieldElement, not InvalidLen
        for i, b in enumerate(blobs):
            if len(b) != BYTES_PER_BLOB:
                if any(_has_noncanonical_element(_buf(e)) for e in blobs[:i]):
                    raise KzgError(BlobError("InvalidFieldElement"))
                raise KzgError(BlobError("InvalidLen"))
        for c in list(commitments) + list(proofs):
            if len(c) != 48:
                raise KzgError(BlsError(ECGroupError("InvalidEncoding")))
        ok = ctypes.c_int32(0)
        rc = self._lib.kzg_verify_blob_proof_batch(
            self._h, b"".join(_buf(b) for b in blobs), b"".join(_buf(c) for c in commitments), b"".join(_buf(p) for p in proofs), n, ctypes.byref(ok)
        )
        self._check(rc, "kzg_verify_blob_proof_batch")
        if rc > 0:
            raise _kzg_error(rc)
        return bool(ok.value)

    def verify_blob_proof_batch_host(self, blobs, commitments, proofs, n: int) -> bool:
        """kzg_verify_blob_proof_batch on n CONTIGUOUS items in host memory: bytes-like objects or raw host addresses
        (ints, e.g. the data_ptr() of a pinned tensor -- pinned memory crosses PCIe at the full rate)."""
        ok = ctypes.c_int32(0)
        args = [a if isinstance(a, int) else _buf(a) for a in (blobs, commitments, proofs)]
        rc = self._lib.kzg_verify_blob_proof_batch(self._h, args[0], args[1], args[2], n, ctypes.byref(ok))
        self._check(rc, "kzg_verify_blob_proof_batch")
        if rc > 0:
            raise _kzg_error(rc)
        return bool(ok.value)

    # -- device-resident entry points (raw HIP pointers, e.g. torch.Tensor.data_ptr()) ---
    def blob_to_commitment_batch_dev(self, d_blobs: int, n: int, d_out48: int, d_status: int, stream: int = 0):
        rc = self._lib.kzg_blob_to_commitment_batch_dev(self._h, d_blobs, n, d_out48, d_status, stream)
        self._check(rc, "kzg_blob_to_commitment_batch_dev")

    def compute_blob_proof_batch_dev(self, d_blobs: int, d_commitments: int, n: int, d_out48: int, d_status: int, stream: int = 0):
        rc = self._lib.kzg_compute_blob_proof_batch_dev(self._h, d_blobs, d_commitments, n, d_out48, d_status, stream)
        self._check(rc, "kzg_compute_blob_proof_batch_dev")

    def verify_blob_proof_batch_dev(self, d_blobs: int, d_commitments: int, d_proofs: int, n: int, stream: int = 0) -> bool:
        ok = ctypes.c_int32(0)
        rc = self._lib.kzg_verify_blob_proof_batch_dev(self._h, d_blobs, d_commitments, d_proofs, n, ctypes.byref(ok), stream)
        self._check(rc, "kzg_verify_blob_proof_batch_dev")
        if rc > 0:
            raise _kzg_error(rc)
        return bool(ok.value)

    # -- device-resident SHARDED calls on a group context: one entry per member, member k's buffers resident on member k's GPU ---
    def _per_member(self, values, what):
        m = self.members
        if len(values) != m:
            raise ValueError("%s: %d entries for a context of %d members" % (what, len(values), m))
        return (ctypes.c_void_p * m)(*[int(v) if v else None for v in values])

    def _streams(self, streams):
        return self._per_member(streams, "streams") if streams is not None else None

    def _counts(self, n_local):
        if len(n_local) != self.members:
            raise ValueError("n_local: %d entries for a context of %d members" % (len(n_local), self.members))
        return (ctypes.c_uint64 * len(n_local))(*n_local)

    def blob_to_commitment_batch_group_dev(self, d_blobs, n_local, d_out48, d_status, streams=None):
        """enqueues on every member and returns without synchronising (like the *_dev calls)"""
        rc = self._lib.kzg_blob_to_commitment_batch_group_dev(self._h, self._per_member(d_blobs, "d_blobs"), self._counts(n_local), self._per_member(d_out48, "d_out48"),
                                                              self._per_member(d_status, "d_status"), self._streams(streams))
        self._check(rc, "kzg_blob_to_commitment_batch_group_dev")

    def compute_blob_proof_batch_group_dev(self, d_blobs, d_commitments, n_local, d_out48, d_status, streams=None):
        rc = self._lib.kzg_compute_blob_proof_batch_group_dev(self._h, self._per_member(d_blobs, "d_blobs"), self._per_member(d_commitments, "d_commitments"),
                                                              self._counts(n_local), self._per_member(d_out48, "d_out48"), self._per_member(d_status, "d_status"),
                                                              self._streams(streams))
        self._check(rc, "kzg_compute_blob_proof_batch_group_dev")

    def verify_blob_proof_batch_group_dev(self, d_blobs, d_commitments, d_proofs, n_local, streams=None) -> bool:
        """Setup::verify_blob_proof_batch over the members' resident shares (global order = member order): the boolean, or the
        reference's first error"""
        ok = ctypes.c_int32(0)
        rc = self._lib.kzg_verify_blob_proof_batch_group_dev(self._h, self._per_member(d_blobs, "d_blobs"), self._per_member(d_commitments, "d_commitments"),
                                                             self._per_member(d_proofs, "d_proofs"), self._counts(n_local), ctypes.byref(ok), self._streams(streams))
        self._check(rc, "kzg_verify_blob_proof_batch_group_dev")
        if rc > 0:
            raise _kzg_error(rc)
        return bool(ok.value)

    def verify_phase1_dev(self, d_blobs: int, d_commitments: int, d_proofs: int, n_local: int, stream: int = 0):
        """-> (session handle, 32-byte transcript root, err6)."""
        root = ctypes.create_string_buffer(32)
        err = (ctypes.c_int32 * 6)()
        sess = ctypes.c_void_p()
        rc = self._lib.kzg_verify_phase1_dev(self._h, d_blobs, d_commitments, d_proofs, n_local, ctypes.cast(root, ctypes.c_void_p), err, ctypes.byref(sess), stream)
        self._check(rc, "kzg_verify_phase1_dev")
        return sess, root.raw, list(err)

    def verify_phase2_dev(self, session, roots: bytes, first_index: int, n_total: int) -> bytes:
        out = ctypes.create_string_buffer(192)
        rc = self._lib.kzg_verify_phase2_dev(session, _buf(roots), len(roots) // 32, first_index, n_total, ctypes.cast(out, ctypes.c_void_p))
        self._check(rc, "kzg_verify_phase2_dev")
        return out.raw

    def verify_session_zy(self, session, first: int, count: int):
        """(z bytes, y bytes) of items [first, first+count) of a phase-1 session, 32 B big-endian each"""
        z = ctypes.create_string_buffer(32 * count)
        y = ctypes.create_string_buffer(32 * count)
        rc = self._lib.kzg_verify_session_zy(session, first, count, ctypes.cast(z, ctypes.c_void_p), ctypes.cast(y, ctypes.c_void_p))
        self._check(rc, "kzg_verify_session_zy")
        return z.raw, y.raw

    def verify_session_destroy(self, session):
        self._lib.kzg_verify_session_destroy(session)

    def verify_batch_finish(self, partials: bytes) -> bool:
        ok = ctypes.c_int32(0)
        rc = self._lib.kzg_verify_batch_finish(self._h, _buf(partials), len(partials) // 192, ctypes.byref(ok))
        self._check(rc, "kzg_verify_batch_finish")
        return bool(ok.value)

    def synth_blobs_dev(self, seed: int, first_index: int, n: int, d_blobs: int, stream: int = 0):
        rc = self._lib.kzg_synth_blobs_dev(self._h, seed, first_index, n, d_blobs, stream)
        self._check(rc, "kzg_synth_blobs_dev")

    def profile_begin(self):
        self._check(self._lib.kzg_profile_begin(self._h), "kzg_profile_begin")

    PROF_KINDS = 8  # KZG_PROF_KINDS

    def profile_end(self) -> dict:
        """HIP-event kernel times since profile_begin: {"msm_ms", "msm_launches", "adds_per_blob", "kinds": {name: (ms, launches)}}."""
        ms = (ctypes.c_double * self.PROF_KINDS)()
        cnt = (ctypes.c_uint64 * self.PROF_KINDS)()
        self._check(self._lib.kzg_profile_end_kinds(self._h, ms, cnt), "kzg_profile_end_kinds")
        kinds = {self._lib.kzg_profile_kind_name(k).decode(): (ms[k], cnt[k]) for k in range(self.PROF_KINDS)}
        kinds[self.msm_kernel_name] = kinds.pop(self._lib.kzg_profile_kind_name(0).decode())  # the fixed-base MSM kernel this context runs
        return {"msm_ms": ms[0], "msm_launches": cnt[0], "adds_per_blob": self._lib.kzg_ctx_adds_per_blob(self._h), "kinds": kinds}

    def selftest_field_mul(self, lanes: int, iters: int) -> int:
        bad = ctypes.c_uint64(0)
        self._check(self._lib.kzg_selftest_field_mul(self._h, lanes, iters, ctypes.byref(bad)), "kzg_selftest_field_mul")
        return bad.value

    def microbench_valu_issue(self, waves_per_simd: int = 2, iters: int = 20000):
        """(SIMD cycles per wave-instruction of v_mad_u64_u32 at `waves_per_simd`, shader clock in GHz under that load)"""
        cyc, ghz = ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._lib.kzg_microbench_valu_issue(self._h, waves_per_simd, iters, ctypes.byref(cyc), ctypes.byref(ghz)), "kzg_microbench_valu_issue")
        return cyc.value, ghz.value

    def clock_probe_launch(self, duration_us: int):
        """eight sleeping probe waves on the context's side stream compare the shader clock with real time for `duration_us`"""
        self._check(self._lib.kzg_clock_probe_launch(self._h, duration_us), "kzg_clock_probe_launch")

    def clock_probe_read(self):
        """(mean, lowest, highest) XCD shader clock in GHz seen by the last probe"""
        a, b, c = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._lib.kzg_clock_probe_read(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "kzg_clock_probe_read")
        return a.value, b.value, c.value

    def microbench_fp_mul(self, lanes: int, iters: int) -> float:
        ms = ctypes.c_float(0)
        rc = self._lib.kzg_microbench_fp_mul(self._h, lanes, iters, ctypes.byref(ms))
        self._check(rc, "kzg_microbench_fp_mul")
        return ms.value
