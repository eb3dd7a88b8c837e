"""CPU-only checks of the drop-in boundary: the C-ABI shared library loads
without a GPU, exports every symbol include/kateth_amd.h declares (and nothing
the header does not), and fails LOUDLY -- no CPU fallback -- when asked to
compute on a machine without a device."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kateth_amd.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from kateth_amd import kzg

    if not os.path.exists(kzg.library_path()):
        g.build_engine()  # hipcc cross-compiles gfx950 here; no GPU needed (the driver's build() normally did this already)

    return kzg.load_library()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kzg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree(lib):
    from kateth_amd import kzg

    declared = declared_functions()
    assert "kzg_blob_to_commitment_batch" in declared and "kzg_verify_blob_proof_batch" in declared
    out = subprocess.check_output(["nm", "-D", "--defined-only", kzg.library_path()], text=True)
    exported = sorted(set(re.findall(r"\bT (kzg_[a-z0-9_]+)\b", out)))
    assert exported == declared, (set(declared) ^ set(exported))
    # the Python mirror binds exactly the same set
    assert sorted(kzg.EXPORTED_SYMBOLS) == declared


def test_no_torch_types_in_signatures():
    text = open(HEADER).read()
    assert "torch" not in text.lower() and "tensor" not in text.lower()
    assert 'extern "C"' in text


def test_header_compiles_as_c_and_links(lib, tmp_path):
    """include/kateth_amd.h is a C header: a C translation unit (gcc -std=c99 -pedantic) that takes the address of every
    declared entry point compiles, and links against the library (resolving every symbol)."""
    from kateth_amd import kzg

    names = declared_functions()
    src = tmp_path / "use_header.c"
    src.write_text('#include "kateth_amd.h"\n#include <stdio.h>\ntypedef void (*fn)(void);\nint main(void) {\n  fn table[] = {%s};\n'
                   '  kzg_config cfg = KZG_CONFIG_INIT;\n  cfg.window_bits = 8;\n  printf("%%d %%d\\n", (int)(sizeof table / sizeof table[0]), (int)cfg.window_bits + KZG_BYTES_PER_G1);\n  return table[0] == 0;\n}\n'
                   % ", ".join("(fn)%s" % n for n in names))
    exe = str(tmp_path / "use_header")
    hip = "/opt/rocm/lib/libamdhip64.so"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-Wno-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe, kzg.library_path(), hip,
                           "-Wl,-rpath," + os.path.dirname(kzg.library_path()), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([exe], text=True).split()
    assert int(out[0]) == len(names) and int(out[1]) == 8 + 48


def test_fails_loudly_without_gpu(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from kateth_amd import kzg

    cfg = kzg._Config.new(0, 8, 0, 0)
    out = ctypes.c_void_p()
    rc = lib.kzg_ctx_create(bytes(4096 * 48), bytes(65 * 96), ctypes.byref(cfg), ctypes.byref(out))
    assert rc == -3 and not out.value  # KZG_FAIL_NO_DEVICE
    assert b"no CPU fallback" in lib.kzg_last_error()
    import kateth_amd

    with pytest.raises(kateth_amd.kzg.EngineError):
        kateth_amd.Setup.load_json(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json"))


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under kateth_amd/ may reference it"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kateth_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".hpp", ".inc", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                hits = re.findall(r"(?:from|import)\s+oracle|oracle[/\\]|oracle\.(?:pyref|cport)|libkzg_cport|pyref", text)
                assert not hits, (os.path.join(dirpath, f), hits)


def test_load_setup_errors_mirror_reference(tmp_path):
    import json

    import kateth_amd

    raw = json.load(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")))
    with pytest.raises(kateth_amd.LoadSetupError, match="Io"):
        kateth_amd.Setup.load_json(str(tmp_path / "missing.json"))
    p = tmp_path / "bad.json"
    p.write_text("{not json")
    with pytest.raises(kateth_amd.LoadSetupError, match="Serde"):
        kateth_amd.Setup.load_json(str(p))
    p.write_text(json.dumps({"g1_lagrange": raw["g1_lagrange"][:-1], "g2_monomial": raw["g2_monomial"]}))
    with pytest.raises(kateth_amd.LoadSetupError, match="InvalidLenG1Lagrange"):  # src/kzg/setup.rs:52-54
        kateth_amd.Setup.load_json(str(p))
    p.write_text(json.dumps({"g1_lagrange": raw["g1_lagrange"], "g2_monomial": raw["g2_monomial"][:3]}))
    with pytest.raises(kateth_amd.LoadSetupError, match="InvalidLenG2Monomial"):  # src/kzg/setup.rs:55-57
        kateth_amd.Setup.load_json(str(p))


def _build_cpp_example(tmp_path):
    from kateth_amd import kzg

    exe = str(tmp_path / "use_kateth_hpp")
    src = os.path.join(ROOT, "tests", "hostcpp", "use_kateth_hpp.cpp")
    hip = "/opt/rocm/lib/libamdhip64.so"
    subprocess.check_call(["g++", "-std=c++17", "-O1", src, "-o", exe, kzg.library_path(), hip, "-Wl,-rpath," + os.path.dirname(kzg.library_path()), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_host_mirror_links_and_fails_loudly_without_gpu(lib, tmp_path):
    """kateth_amd/host/kateth.hpp (the compiled-language mirror of kateth's Setup API) compiles, links
    against the C ABI + the system HIP runtime, and -- with no GPU -- reports KZG_FAIL_NO_DEVICE."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present (covered by the gpu test)")
    exe = _build_cpp_example(tmp_path)
    rc = subprocess.call([exe])
    assert rc == 42


def test_cpp_mirror_p1_compress_and_blob(lib, tmp_path):
    """kateth::P1::compress (the caller-side `Compress::compress`, src/bls.rs:491-503, on the 96-byte blst_p1_affine image the
    *_affine entry points return) against the oracle; kateth::Blob::{random, from_slice} (src/blob.rs:26-37,66-76)"""
    import random

    from kateth_amd import kzg
    from oracle.pyref import bls

    exe = str(tmp_path / "p1_compress")
    hip = "/opt/rocm/lib/libamdhip64.so"
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "hostcpp", "p1_compress.cpp"), "-o", exe, kzg.library_path(), hip,
                           "-Wl,-rpath," + os.path.dirname(kzg.library_path()), "-Wl,-rpath,/opt/rocm/lib"])
    rnd = random.Random(3)
    for k in range(6):
        pt = bls.g1_mul(bls.G1_GEN, rnd.randrange(1, bls.R))
        if k % 2:
            pt = bls.g1_neg(pt)
        x, y = pt
        img = (x * (1 << 384) % bls.P).to_bytes(48, "little") + (y * (1 << 384) % bls.P).to_bytes(48, "little")
        assert subprocess.check_output([exe, img.hex()], text=True).strip() == bls.g1_compress(pt).hex()
    assert subprocess.check_output([exe, bytes(96).hex()], text=True).strip() == bls.g1_compress(None).hex()
    assert subprocess.check_output([exe, "blob"], text=True).strip() == "131072"


@pytest.mark.gpu
def test_cpp_host_mirror_round_trip_on_gpu(tmp_path):
    import json

    raw = json.load(open(os.path.join(ROOT, "tests", "golden", "trusted_setup_4096.json")))
    g1 = tmp_path / "g1.bin"
    g2 = tmp_path / "g2.bin"
    g1.write_bytes(b"".join(bytes.fromhex(s[2:]) for s in raw["g1_lagrange"]))
    g2.write_bytes(b"".join(bytes.fromhex(s[2:]) for s in raw["g2_monomial"]))
    exe = _build_cpp_example(tmp_path)
    out = subprocess.run([exe, str(g1), str(g2)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "verify=1" in out.stdout


def header_table_rows():
    """rows of the table-class table in include/kateth_amd.h's kzg_config comment: class -> (blocks per 64 points,
    entries per 64 points and group, default G, stated resident bytes, additions per blob)"""
    rows = {}
    for line in open(HEADER).read().splitlines():
        m = re.match(r"\s*\*\s+(\d+)\s+\|([^|]+)\|([^|]+)\|\s*(\d+)[^|]*\|([^|]+)\|\s*([\d,]+)\s*$", line)
        if not m:
            continue
        cls, blocks, e_txt, g, size_txt, adds = m.groups()
        e = eval(e_txt.strip().replace("^", "**"), {"__builtins__": {}})  # "4 * 2^15"
        nblocks = blocks.count("+") + 1 if "+" in blocks else int(blocks.split("x")[0])
        val, unit = re.match(r"\s*([\d.]+)\s*(GiB|GB|MB)", size_txt).groups()
        rows[int(cls)] = (nblocks, e, int(g), float(val) * {"GiB": 2**30, "GB": 1e9, "MB": 1e6}[unit], int(adds.replace(",", "")))
    return rows


def test_header_table_sizes_follow_the_geometry():
    """the memory a maintainer provisions from the public header must be what the engine allocates: stated bytes =
    G * 64 * e * 96 within rounding, additions per blob = 256 planes x blocks x 64 (tests/test_gpu_parity.py compares the same
    rows with kzg_ctx_table_bytes / kzg_ctx_adds_per_blob on the device)"""
    rows = header_table_rows()
    assert sorted(rows) == [4, 8, 16, 22]
    for cls, (nblocks, e, g, stated, adds) in rows.items():
        assert abs(g * 64 * e * 96 - stated) / stated < 0.01, cls
        assert adds == 256 * nblocks * 64, cls
    assert rows[22][:3] == (3, 1 << 22, 8) and rows[16][:3] == (4, 4 << 15, 16)


def test_product_sources_have_one_msm_path():
    """the window-table cross-check kernels live under tests/window_msm and hook in through MsmOverride; the product sources
    carry no conditional compilation for them and the product library none of their code"""
    from kateth_amd import kzg

    csrc = os.path.join(ROOT, "kateth_amd", "csrc")
    for name in os.listdir(csrc):
        if name.endswith((".hip", ".cuh", ".hpp", ".inc", ".h")):
            text = open(os.path.join(csrc, name)).read()
            assert "KZG_TEST_WINDOW_MSM" not in text and "k_msm_fixed28" not in text, name
    syms = subprocess.check_output(["strings", kzg.library_path()], text=True)
    assert "k_msm_comb30" in syms and "k_msm_fixed" not in syms and "kzg_test_read_wave_times" not in syms


def test_library_load_leaves_the_environment_alone(lib):
    """round 5: no load-time constructor calls setenv any more -- GPU_MAX_HW_QUEUES is the host program's to set;
    the library only NAMES the setting (kzg_recommended_env) and its undefined-symbol list has no setenv / putenv"""
    import sys

    from kateth_amd import kzg

    LIB = kzg.library_path()
    # the library binds to the HIP runtime already in the process (no DT_NEEDED): load one first, as a C host program's link line does
    prog = ("import ctypes, os, sys; ctypes.CDLL('/opt/rocm/lib/libamdhip64.so', mode=ctypes.RTLD_GLOBAL); lib = ctypes.CDLL(sys.argv[1]); "
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; lib.kzg_recommended_env.restype = ctypes.c_char_p; "
            "print((libc.getenv(b'GPU_MAX_HW_QUEUES') or b'unset').decode(), lib.kzg_recommended_env().decode())")
    base = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    got = subprocess.run([sys.executable, "-c", prog, LIB], env=base, capture_output=True, text=True, timeout=120)
    assert got.returncode == 0 and got.stdout.split() == ["unset", "GPU_MAX_HW_QUEUES=16"], (got.stdout, got.stderr[-300:])
    got = subprocess.run([sys.executable, "-c", prog, LIB], env=dict(base, GPU_MAX_HW_QUEUES="4"), capture_output=True, text=True, timeout=120)
    assert got.returncode == 0 and got.stdout.split() == ["4", "GPU_MAX_HW_QUEUES=16"], (got.stdout, got.stderr[-300:])
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", LIB], text=True)
    assert " setenv" not in undefined and " putenv" not in undefined
    # the Python package (which owns ITS process) still defaults the variable before anything initialises HIP
    got = subprocess.run([sys.executable, "-c", "import os, kateth_amd; print(os.environ.get('GPU_MAX_HW_QUEUES'))"], env=base, capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert got.returncode == 0 and got.stdout.strip() == "16", (got.stdout, got.stderr[-300:])


def test_every_kernel_is_compiled_once(lib):
    """round 5: every kernel header is included by exactly one translation unit, which exports host launchers
    (engine_internal.hpp) -- so no gfx950 kernel symbol appears in two of the library's objects (rounds 1-4: every kernel of
    blob_kernels / msm_* once per engine*.hip, an 8.9-MB library; now 4.6 MB)"""
    import __graft_entry__ as g

    bundler, readelf = "/opt/rocm/lib/llvm/bin/clang-offload-bundler", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(bundler) and os.path.exists(readelf)):
        pytest.skip("ROCm LLVM tools not present")
    seen = {}
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        for unit in g.ENGINE_UNITS:
            obj = os.path.join(g.CSRC, unit + ".o")
            assert os.path.exists(obj), "build() leaves the objects in-tree"
            fb, co = os.path.join(tmp, unit + ".fb"), os.path.join(tmp, unit + ".co")
            subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb])
            if os.path.getsize(fb) == 0:
                continue  # a unit without device code (engine_multi)
            subprocess.check_call([bundler, "--unbundle", "--type=o", "--input=" + fb, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
            notes = subprocess.check_output([readelf, "--notes", co], text=True)
            for sym in re.findall(r"\.symbol:\s*'?([\w.$@]+?)\.kd'?\s", notes):
                seen.setdefault(sym, []).append(unit)
    assert len(seen) >= 45, sorted(seen)
    twice = {k: v for k, v in seen.items() if len(v) > 1}
    assert not twice, twice
    assert any("k_msm_comb30" in k for k in seen) and any("k_challenge_pair" in k for k in seen) and any("k_eval_frac" in k for k in seen)


def test_no_exception_leaves_an_entry_point(lib):
    """'never unwind across the boundary'.  Every extern "C" function that returns int32_t and has a body of its own is
    a function-try-block whose handler turns what was thrown into KZG_FAIL_HOST (abi_exception); the test hook throws a
    std::bad_alloc, a std::runtime_error and an int inside one such function (no GPU involved)."""
    import ctypes

    csrc = os.path.join(ROOT, "kateth_amd", "csrc")
    checked = 0
    for name in sorted(os.listdir(csrc)):
        if not (name.startswith("engine") and name.endswith(".hip")):
            continue
        text = open(os.path.join(csrc, name)).read()
        for m in re.finditer(r'^extern "C" int32_t (kzg_\w+)\(([^{};]*)\)\s*(try\s*)?\{[^\n]*$', text, flags=re.M):
            if m.group(0).rstrip().endswith("}"):
                continue  # a one-line getter: nothing in it can throw
            assert m.group(3), "%s: %s is not a function-try-block" % (name, m.group(1))
            checked += 1
    assert checked >= 38
    assert text.count("abi_exception()") >= 1
    lib.kzg_last_error.restype = ctypes.c_char_p
    for kind, needle in ((0, b"bad_alloc"), (1, b"runtime_error inside an entry point"), (2, b"unexpected C++ exception")):
        assert lib.kzg_selftest_exception_guard(kind) == -7
        assert needle in lib.kzg_last_error(), lib.kzg_last_error()
    assert lib.kzg_selftest_exception_guard(3) == 0
