#!/usr/bin/env python3
"""Instruction mix per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only)."""
import collections
import re
import sys


def main(path, detail=()):
    s = open(path).read()
    for name in re.findall(r"^(_Z\w+):", s, flags=re.M):
        a = s.index("\n" + name + ":")
        b = s.find(".Lfunc_end", a)  # closes kernels and device functions alike
        body = s[a:b if b > 0 else len(s)]
        ins = []
        for l in body.split("\n"):
            t = l.strip()
            if not l.startswith("\t") or not t or t[0] in ".;":
                continue
            ins.append(t.split()[0])
        c = collections.Counter(ins)
        scratch = sum(v for k, v in c.items() if "scratch" in k)
        print("%-60s %6d instr  mad_u64 %5d  scratch %4d  s_nop %4d" % (name[:60], len(ins), c["v_mad_u64_u32"], scratch, c["s_nop"]))
        if any(d in name for d in detail):
            print("    ", c.most_common(18))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
