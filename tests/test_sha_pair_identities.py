"""The lane-pair SHA-256 rounds (kateth_amd/csrc/sha256.cuh, "the 64 rounds on a PAIR of lanes"; the rounds themselves are generated assembly:
kateth_amd/csrc/sha_pair_asm.cuh, tools/gen_sha_pair_asm.py) are device-only (DPP lane exchange) and are covered end to end by the
GPU parity tests.  Here, on the CPU:
  * Maj(a, b, c) = Ch(~(a ^ b), b, c);
  * the selector v_bitop3_b32(a0, a1, role, 0xD2) is a0 for role = 0 and ~(a0 ^ a1) for role = all-ones;
  * the GENERATED INSTRUCTION STREAM itself is interpreted on a group of eight lanes (four X / Y pairs: DPP row_half_mirror and
    the identity permutation with their bank masks, the LDS quads, the lgkmcnt waits) and must give the FIPS 180-4 compression
    function for four independent messages at once; the interpreter also checks the two hazards the stream covers by instruction
    order alone -- a DPP read at least two instructions after the VALU write of its source, no W + K register read before the
    wait that covers its load;
  * the thread -> (role, blob slot) map of the consumer waves is a bijection onto 64 slots x two roles with partners j <-> 7 - j."""
import hashlib
import os
import re
import random
import struct

import pytest

M = 0xFFFFFFFF
K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
H0 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


def rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & M


def ch(e, f, g):
    return ((e & f) ^ (~e & g)) & M


def maj(a, b, c):
    return ((a & b) ^ (a & c) ^ (b & c)) & M


def bitop3(a, b, c, table):
    """v_bitop3_b32: result bit = table[(a_bit << 2) | (b_bit << 1) | c_bit]"""
    out = 0
    for i in range(32):
        idx = (((a >> i) & 1) << 2) | (((b >> i) & 1) << 1) | ((c >> i) & 1)
        out |= ((table >> idx) & 1) << i
    return out


def test_maj_is_ch_with_a_xor_b_selector():
    rnd = random.Random(1)
    for _ in range(2000):
        a, b, c = (rnd.getrandbits(32) for _ in range(3))
        assert maj(a, b, c) == ch(~(a ^ b) & M, b, c)


def test_selector_truth_table_0xD2():
    rnd = random.Random(2)
    for _ in range(500):
        a0, a1 = rnd.getrandbits(32), rnd.getrandbits(32)
        assert bitop3(a0, a1, 0, 0xD2) == a0
        assert bitop3(a0, a1, M, 0xD2) == (~(a0 ^ a1)) & M
    assert bitop3(0xF0F0F0F0, 0xCCCCCCCC, 0xAAAAAAAA, 0xCA) == ch(0xF0F0F0F0, 0xCCCCCCCC, 0xAAAAAAAA)  # the Ch table the kernel uses


def schedule(block16):
    w = list(block16)
    for i in range(16, 64):
        s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)
        s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10)
        w.append((w[i - 16] + s0 + w[i - 7] + s1) & M)
    return w


ASM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kateth_amd", "csrc", "sha_pair_asm.cuh")
LANES = 8  # one half row: lanes 0-3 are DPP bank 0 (X), lanes 4-7 bank 1 (Y); lane j's partner is 7 - j
ROW_QUADS = 65  # a schedule row: 64 slots + the zero quad of the Y lanes
BLOCK_WORDS = 16 * ROW_QUADS * 4


def asm_lines(nb):
    """the instruction lines of sha256_blocks_pair_asm<nb>"""
    text = open(ASM).read()
    start = text.index("void sha256_blocks_pair_asm%d(" % nb)
    body = text[start:text.index('      : "+v"(a0)', start)]
    return [m.group(1) for m in re.finditer(r'^\s*"([^"]+?)\\n\\t"', body, re.M)]


class Wave:
    """registers as per-lane lists; interprets exactly the opcodes the generator emits"""

    def __init__(self, operands):
        self.reg = dict(operands)
        self.written_at = {}  # register -> index of the instruction slot (VALU or s_nop wait states) that wrote it
        self.slot = 0
        self.loads = []  # destination register quads in issue order
        self.complete = 0  # loads [0, complete) have returned

    def get(self, name):
        if name not in self.reg:
            raise AssertionError("read of unwritten register " + name)
        return self.reg[name]

    def dpp(self, src, ctrl):
        v = self.get(src)
        assert self.slot - self.written_at.get(src, -10) >= 3, "DPP read of %s too close to its write" % src
        if ctrl == "row_half_mirror":
            return [v[7 - j] for j in range(LANES)]
        assert ctrl == "quad_perm:[0,1,2,3]"
        return list(v)

    def put(self, name, values, banks=0xF):
        old = self.reg.get(name, [None] * LANES)
        self.reg[name] = [values[j] if (banks >> (j // 4)) & 1 else old[j] for j in range(LANES)]
        self.written_at[name] = self.slot

    def check_loaded(self, name):
        for k, quad in enumerate(self.loads):
            if name in quad:
                assert k < self.complete, "%s read before the wait that covers its load" % name

    def run(self, lines, lds):
        for line in lines:
            op, rest = line.split(None, 1)
            if op == "ds_read_b128":
                m = re.match(r"v\[(\d+):(\d+)\], (\S+) offset:(\d+)", rest)
                lo, hi, addr, off = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4))
                assert hi == lo + 3
                base = self.get(addr)
                for q in range(4):
                    self.reg["v%d" % (lo + q)] = [lds[(base[j] + off) // 4 + q] for j in range(LANES)]
                self.loads.append(["v%d" % (lo + q) for q in range(4)])
                continue
            if op == "s_waitcnt":
                n = int(re.match(r"lgkmcnt\((\d+)\)", rest).group(1))
                self.complete = max(self.complete, len(self.loads) - n)
                continue
            if op == "s_nop":
                self.slot += int(rest) + 1
                continue
            args = [x.strip() for x in rest.split(",")]
            if op == "v_add_u32_dpp":
                dst, src0 = args[0], args[1]
                src1, ctrl, rm, bm = args[2].split(None, 1)[0], None, None, None
                tail = args[2].split(None, 1)[1] + ("," + ",".join(args[3:]) if len(args) > 3 else "")
                ctrl = "row_half_mirror" if "row_half_mirror" in tail else re.search(r"quad_perm:\[[\d,]+\]", tail).group(0)
                assert "row_mask:0xf" in tail
                banks = int(re.search(r"bank_mask:0x([0-9a-f])", tail).group(1), 16)
                a, b = self.dpp(src0, ctrl), self.get(src1)
                for r in (src0, src1):
                    self.check_loaded(r)
                self.put(dst, [(a[j] + b[j]) & M if a[j] is not None and b[j] is not None else None for j in range(LANES)], banks)
            elif op == "v_alignbit_b32":
                d, x, y, k = args
                assert x == y
                xs, ks = self.get(x), self.get(k)
                self.put(d, [rotr(xs[j], ks[j]) for j in range(LANES)])
            elif op == "v_bitop3_b32":
                d, x, y = args[0], args[1], args[2]
                z, table = args[3].split()[0], int(args[3].split("bitop3:")[1], 16)
                xs, ys, zs = self.get(x), self.get(y), self.get(z)
                self.put(d, [bitop3(xs[j], ys[j], zs[j], table) for j in range(LANES)])
            elif op == "v_add3_u32":
                d, x, y, z = args
                self.check_loaded(z)
                xs, ys, zs = self.get(x), self.get(y), self.get(z)
                self.put(d, [(xs[j] + ys[j] + zs[j]) & M for j in range(LANES)])
            elif op == "v_add_u32_e32":
                d, x, y = args
                xs, ys = self.get(x), self.get(y)
                self.put(d, [(xs[j] + ys[j]) & M for j in range(LANES)])
            else:
                raise AssertionError("opcode the interpreter does not know: " + op)
            self.slot += 1


def run_statement(nb, state, blocks_per_slot):
    """one generated statement over nb blocks: state[slot] = 8 chaining words, blocks_per_slot[slot] = nb lists of 16 message words"""
    lines = asm_lines(nb)
    assert lines[0].startswith("ds_read_b128") and len(lines) > 600 * nb
    is_y = [j >= 4 for j in range(LANES)]
    slot = [7 - j if is_y[j] else j for j in range(LANES)]
    lds = [0] * (nb * BLOCK_WORDS)
    for s_ in range(4):
        for blk in range(nb):
            w = schedule(blocks_per_slot[s_][blk])
            for t in range(64):
                lds[blk * BLOCK_WORDS + ((t // 4) * ROW_QUADS + s_) * 4 + (t % 4)] = (w[t] + K[t]) & M
    ops = {"%%%d" % q: [state[slot[j]][q] if is_y[j] else state[slot[j]][4 + q] for j in range(LANES)] for q in range(4)}
    for blk in range(nb):
        ops["%%%d" % (4 + blk)] = [4 * (blk * BLOCK_WORDS + 4 * (64 if is_y[j] else slot[j])) for j in range(LANES)]
    for q, (x, y) in enumerate(((6, 2), (11, 13), (25, 22))):
        ops["%%%d" % (4 + nb + q)] = [y if is_y[j] else x for j in range(LANES)]
    ops["%%%d" % (7 + nb)] = [M if is_y[j] else 0 for j in range(LANES)]
    wave = Wave(ops)
    wave.run(lines, lds)
    assert wave.complete == len(wave.loads), "reads still outstanding at the end of the statement"
    for j in range(LANES):
        for q in range(4):
            state[slot[j]][q if is_y[j] else 4 + q] = wave.reg["%%%d" % q][j]


@pytest.mark.parametrize("nb", [1, 2, 4])
def test_generated_blocks_equal_sha256_on_four_lane_pairs(nb):
    rnd = random.Random(3 + nb)
    # four messages of 8 blocks after padding (448 + 55 bytes and shorter ones padded with whole zero blocks is not SHA: so four
    # messages of the SAME padded length, different content)
    msgs = [bytes(rnd.getrandbits(8) for _ in range(64 * 8 - 9 - k)) for k in range(4)]
    padded = [m + b"\x80" + b"\x00" * ((55 - len(m)) % 64) + struct.pack(">Q", 8 * len(m)) for m in msgs]
    assert all(len(p) == 64 * 8 for p in padded)
    state = [list(H0) for _ in msgs]
    for first in range(0, 8, nb):
        run_statement(nb, state, [[struct.unpack(">16I", p[64 * (first + k):64 * (first + k) + 64]) for k in range(nb)] for p in padded])
    for s_, m in enumerate(msgs):
        assert struct.pack(">8I", *state[s_]) == hashlib.sha256(m).digest()


def test_consumer_thread_map():
    """sha_pair_is_y / sha_pair_slot (sha256.cuh): 128 consumer threads -> 64 blob slots x {X, Y}; partners are the lanes j and
    7 - j of a group of eight; X lanes sit in DPP banks 0 and 2, Y lanes in banks 1 and 3"""
    seen = set()
    for tid in range(128):
        is_y = (tid & 4) != 0
        slot = (tid >> 3) * 4 + (7 - (tid & 7) if is_y else (tid & 7))
        partner = (tid & ~7) | (7 - (tid & 7))
        p_is_y = (partner & 4) != 0
        p_slot = (partner >> 3) * 4 + (7 - (partner & 7) if p_is_y else (partner & 7))
        assert p_is_y != is_y and p_slot == slot
        assert ((tid % 16) // 4) % 2 == (1 if is_y else 0)
        seen.add((slot, is_y))
    assert seen == {(s_, y) for s_ in range(64) for y in (False, True)}


def test_committed_header_is_what_the_generator_writes():
    """kateth_amd/csrc/sha_pair_asm.cuh is generated (tools/gen_sha_pair_asm.py) and committed: the two must not drift apart"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_sha_pair_asm", os.path.join(os.path.dirname(os.path.dirname(ASM)), "..", "tools", "gen_sha_pair_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    src, counts = gen.render()
    assert src == open(ASM).read()
    assert counts == [len(asm_lines(nb)) for nb in (1, 2, 4)]
