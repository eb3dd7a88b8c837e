#!/usr/bin/env python3
"""Generates kateth_amd/csrc/sha_pair_asm.cuh: SHA-256 blocks on lane pairs (sha256.cuh, "the 64 rounds on a PAIR of lanes") -- the 64 rounds
and the feed-forward of 1, 2 or 4 consecutive blocks as ONE inline-asm statement each.

Why generated assembly: the rounds of a lone wave are bound by its instruction-issue slots (one per ~4.1 cycles, s_nop and
s_waitcnt included), and hipcc pads every asm boundary and every DPP hazard it cannot see through with an s_nop.  Written out
by hand the round is 10 VALU instructions with every hazard covered by the order of the instructions themselves:

  v_alignbit x3, v_bitop3 (selector), v_bitop3 (xor3), v_bitop3 (Ch / Maj), v_add3 (t = Sigma + Ch + hw[i]),
  v_add_u32_dpp hw[i+3] += a0     (identity permutation, X banks only: h of round i+3 is e of round i)
  v_add_u32_dpp n = mirror(a3) + t    (X banks: e' = d + T1 -- d is the partner's a3)
  v_add_u32_dpp n = mirror(t) + t     (Y banks: a' = T1 + T2; t was written three instructions earlier: the DPP read needs two)

and the new value lands in the register of the dead a3, so the four state registers rotate by name, not by moves; the first four
rounds of a block write fresh registers instead, which leaves the chaining value in place for the feed-forward (four adds, no
copies).  The 64 W + K values of a block come from LDS as 16 ds_read_b128 (rows of quads: [t / 4][slot]).  The first block of a
statement waits for them two groups at a time; every FURTHER block's values are read into a second register set during the
first 16 rounds of the block before it, so their latency -- exposed once per statement -- is hidden and one wait suffices."""
import os

WSET = [64, 128]  # v[64:127], v[128:191]: the W + K values of even / odd blocks of a statement (+ h on X lanes after the pre-add)
R1, R2, R3, SEL, T = "v192", "v193", "v194", "v195", "v196"
FRESH = ["v197", "v198", "v199", "v200"]  # the working state from round 4 on; %0..%3 keep the chaining value for the feed-forward
CLOBBERS = ["v%d" % r for r in range(64, 201)]
ROW_BYTES = 65 * 16  # a row of the schedule: 64 slots of one quad + the all-zero quad the Y lanes read


def body(nb):
    """instruction lines of a statement over nb blocks; operands: %0..%3 state, %4..%(3+nb) the lane's LDS address in each block's
    schedule, then k1, k2, k3, ymask"""
    k1, k2, k3, ym = ("%%%d" % (4 + nb + q) for q in range(4))
    out = []
    issued = []  # (block, group) in issue order

    def w(b, i):
        return "v%d" % (WSET[b & 1] + i)

    def load(b, g):
        r = WSET[b & 1] + 4 * g
        out.append("ds_read_b128 v[%d:%d], %%%d offset:%d" % (r, r + 3, 4 + b, ROW_BYTES * g))
        issued.append((b, g))

    def wait_for(b, g):
        # LDS reads return in order: at most this many younger ones may be outstanding (the counter has four bits: a smaller
        # number only waits for reads that were issued even earlier)
        out.append("s_waitcnt lgkmcnt(%d)" % min(15, len(issued) - 1 - issued.index((b, g))))

    xonly = "quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x5"
    for g in range(16):
        load(0, g)
    a = ["%0", "%1", "%2", "%3"]  # a0 a1 a2 a3
    for b in range(nb):
        wait_for(b, 1 if b == 0 else 15)  # first block: groups 0 and 1, then two groups per wait; later blocks: all of it, long landed
        out.append("v_add_u32_dpp %s, %s, %s %s" % (w(b, 0), a[3], w(b, 0), xonly))  # h of rounds 0, 1, 2 = h, g, f
        out.append("v_add_u32_dpp %s, %s, %s %s" % (w(b, 1), a[2], w(b, 1), xonly))
        out.append("v_add_u32_dpp %s, %s, %s %s" % (w(b, 2), a[1], w(b, 2), xonly))
        for i in range(64):
            a0, a1, a2, a3 = a
            # both halves of the new value are written (X banks, then Y banks), so it may land in any register: in a fresh one for
            # the first four rounds -- the chaining value stays where it is -- and in the register of the dead a3 afterwards
            n = FRESH[i] if i < 4 else a3
            out.append("v_alignbit_b32 %s, %s, %s, %s" % (R1, a0, a0, k1))
            out.append("v_alignbit_b32 %s, %s, %s, %s" % (R2, a0, a0, k2))
            out.append("v_alignbit_b32 %s, %s, %s, %s" % (R3, a0, a0, k3))
            out.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0xd2" % (SEL, a0, a1, ym))
            out.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0x96" % (R1, R1, R2, R3))
            out.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0xca" % (SEL, SEL, a1, a2))
            out.append("v_add3_u32 %s, %s, %s, %s" % (T, R1, SEL, w(b, i)))
            if i + 3 < 64:
                if b == 0 and (i + 3) % 8 == 0:
                    wait_for(0, (i + 3) // 4 + 1)
                out.append("v_add_u32_dpp %s, %s, %s %s" % (w(b, i + 3), a0, w(b, i + 3), xonly))
            else:
                out.append("s_nop 0")  # the second wait state between t and its DPP read
            out.append("v_add_u32_dpp %s, %s, %s row_half_mirror row_mask:0xf bank_mask:0x5" % (n, a3, T))
            out.append("v_add_u32_dpp %s, %s, %s row_half_mirror row_mask:0xf bank_mask:0xa" % (n, T, T))
            if b + 1 < nb and i < 16:
                load(b + 1, i)  # the next block's values into the other register set
            a = [n, a0, a1, a2]
        for q in (3, 2, 1, 0):  # feed-forward into the chaining value; a3 first: the next block's pre-adds DPP-read a3, a2, a1
            out.append("v_add_u32_e32 %%%d, %%%d, %s" % (q, q, a[q]))
        a = ["%0", "%1", "%2", "%3"]
    return out


def function(nb):
    lines = body(nb)
    text = "\n".join('      "%s\\n\\t"' % l for l in lines)
    addrs = ", ".join("uint32_t lds%d" % b for b in range(nb))
    ops = ", ".join('"v"(lds%d)' % b for b in range(nb))
    return """// %d block%s: %d instruction slots
__device__ __forceinline__ void sha256_blocks_pair_asm%d(uint32_t& a0, uint32_t& a1, uint32_t& a2, uint32_t& a3, %s, uint32_t k1, uint32_t k2,
                                                         uint32_t k3, uint32_t ymask) {
  asm volatile(
%s
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
      : %s, "v"(k1), "v"(k2), "v"(k3), "v"(ymask)
      : %s, "memory");
}
""" % (nb, "" if nb == 1 else "s", len(lines), nb, addrs, text, ops, ", ".join('"%s"' % c for c in CLOBBERS)), len(lines)


def render():
    """(source text of sha_pair_asm.cuh, instruction slots of the 1 / 2 / 4 block statements)"""
    parts, counts = [], []
    for nb in (1, 2, 4):
        src, n = function(nb)
        parts.append(src)
        counts.append(n)
    src = """// GENERATED by tools/gen_sha_pair_asm.py -- do not edit.
#pragma once

namespace kzg {
#if defined(__HIPCC__)
// Consecutive SHA-256 blocks on a lane pair -- the 64 rounds and the feed-forward into the chaining value a0..a3 of each (roles,
// lane layout and data flow: sha256.cuh, "the 64 rounds on a PAIR of lanes").  ldsB: byte address in LDS of this lane's quad in row 0 of block B's
// schedule -- X lanes: quad `slot` of the [16 rows][65 quads] schedule (row t / 4 holds W[t] + K[t] of four rounds for 64 slots,
// then an all-zero quad), Y lanes: that zero quad; row g at ldsB + %d g.
constexpr uint32_t SHA_PAIR_ROW_QUADS = 65;
%s#endif
}  // namespace kzg
""" % (ROW_BYTES, "\n".join(parts))
    return src, counts


def main():
    src, counts = render()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kateth_amd", "csrc", "sha_pair_asm.cuh")
    open(path, "w").write(src)
    print("wrote", path, counts, "instruction slots for 1 / 2 / 4 blocks")


if __name__ == "__main__":
    main()
