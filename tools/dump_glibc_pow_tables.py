"""Regenerates extpom_amd/csrc/glibc_pow_tables.h from the libm this machine runs (glibc 2.35 x86-64).

The table words are read from libm.so.6's read-only data at the addresses the FMA variant of
__ieee754_pow uses (found by disassembling it: the pow wrapper -> IRELATIVE resolver -> variant);
the addresses below are those of Ubuntu GLIBC 2.35-0ubuntu3.11.  Sanity anchors (ln2hi, ln2lo,
A[0] = -0.5, 2^(0/128) = 1.0) are asserted so that a different build fails instead of producing
a wrong header.  tools/check_glibc_pow_clone.c verifies the restated algorithm against pow().
"""
import struct
import sys

LIBM = "/lib/x86_64-linux-gnu/libm.so.6"
POW_LOG_DATA, EXP_DATA = 0xB1B20, 0xAF960


def main(out):
    f = open(LIBM, "rb").read()
    e_phoff = struct.unpack_from("<Q", f, 32)[0]
    e_phentsize, e_phnum = struct.unpack_from("<HH", f, 54)
    segs = []
    for i in range(e_phnum):
        p = struct.unpack_from("<IIQQQQQQ", f, e_phoff + i * e_phentsize)
        if p[0] == 1:
            segs.append((p[3], p[2], p[5]))

    def rd(v, n):
        for va, off, sz in segs:
            if va <= v < va + sz:
                return f[off + v - va:off + v - va + n]
        raise KeyError(hex(v))

    d = lambda v: struct.unpack("<d", rd(v, 8))[0]
    q = lambda v: struct.unpack("<Q", rd(v, 8))[0]
    L, E = POW_LOG_DATA, EXP_DATA
    assert d(L).hex() == "0x1.62e42fefa3800p-1" and d(L + 8).hex() == "0x1.ef35793c76730p-45", "not the expected libm build"
    assert d(L + 16) == -0.5 and q(E + 0x78) == 0x3FF0000000000000 and d(E + 8) == 0x1.8p52
    o = [open(out).read().split("#pragma once")[0] + "#pragma once"]
    o.append("#define GPOW_LN2HI %s\n#define GPOW_LN2LO %s" % (d(L).hex(), d(L + 8).hex()))
    o.append("#define GPOW_A { " + ", ".join(d(L + 16 + 8 * i).hex() for i in range(7)) + " }")
    o.append("// pow log table: { invc, logc, logctail } for the 128 sub-intervals of [OFF, 2*OFF)")
    o.append("#define GPOW_LOGTAB { \\")
    for i in range(128):
        b = L + 72 + 32 * i
        o.append("  { %s, %s, %s }, \\" % (d(b).hex(), d(b + 16).hex(), d(b + 24).hex()))
    o[-1] = o[-1].rstrip(", \\") + " }"
    o.append("#define GEXP_INVLN2N %s\n#define GEXP_SHIFT %s\n#define GEXP_NEGLN2HIN %s\n#define GEXP_NEGLN2LON %s"
             % tuple(d(E + 8 * i).hex() for i in range(4)))
    o.append("#define GEXP_C { " + ", ".join(d(E + 0x20 + 8 * i).hex() for i in range(4)) + " }   // C2..C5")
    o.append("// exp table: 2^(k/128) as { tail bits, scale bits - (k << 45) }")
    o.append("#define GEXP_TAB { \\")
    for i in range(128):
        o.append("  0x%016xULL, 0x%016xULL, \\" % (q(E + 0x70 + 16 * i), q(E + 0x70 + 16 * i + 8)))
    o[-1] = o[-1].rstrip(", \\") + " }"
    open(out, "w").write("\n".join(o) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "extpom_amd/csrc/glibc_pow_tables.h")
