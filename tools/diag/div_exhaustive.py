#!/usr/bin/env python3
"""Diagnostic: csrc/ca_math.h div_ir / sqrt_ir (the correctly rounded fp32 division / square root without the compiler's range
scaling and fix-up instructions) against the host's IEEE results, on many more operands than the unit tests take.
  python tools/diag/div_exhaustive.py [rounds=256]      (4M operands per round and operation)
Inside the domain (2^-60 <= |a|, |b| <= 2^60 or a = 0; x = 0 or x >= 2^-96) every result must be bit-identical; the script also
REPORTS what happens outside it (denormal operands, quotients that leave the normal range), which is why the callers must be
in range by construction."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from tests import helpers as H

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 256
env = H.make_gpu(1, 4, "crowd", H.scenario_params("crowd", 4), seed=1)
rng = np.random.RandomState(123)
n = 1 << 22


def dbg(op, inp, out):
    env._call("ca_debug_math", env.h, op, inp.ctypes.data, out.ctypes.data, len(out))


def mant():
    return (rng.randint(0, 1 << 23, n).astype(np.uint32) | np.uint32(0x3F800000)).view(np.float32)


def sgn():
    return rng.choice(np.float32([-1, 1]), n)


t0 = time.time()
tested = bad_div = bad_sqrt = 0
for r in range(rounds):
    b = np.ldexp(mant(), rng.randint(-60, 61, n)).astype(np.float32) * sgn()
    kind = r % 4
    if kind == 0:
        a = np.ldexp(mant(), rng.randint(-60, 61, n)).astype(np.float32) * sgn()
    elif kind == 1:   # next to rounding boundaries: a = fl(q b) moved by an ulp
        q = np.ldexp(mant(), rng.randint(-30, 31, n)).astype(np.float32)
        a = np.nextafter((q * b).astype(np.float32), np.float32(np.inf) * sgn()).astype(np.float32)
    elif kind == 2:   # the clip's ranges
        b = np.ldexp(mant(), rng.randint(-17, 1, n)).astype(np.float32) * sgn()
        a = np.ldexp(mant(), rng.randint(-100, 26, n)).astype(np.float32) * sgn()     # (beyond the stated domain: still exact, the quotient stays normal)
    else:             # reciprocals and exact quotients
        a = np.where(rng.randint(0, 2, n) == 0, np.float32(1.0), (b * rng.randint(-1000, 1001, n).astype(np.float32)).astype(np.float32)).astype(np.float32)
    out = np.empty(n, np.float32)
    dbg(6, np.ascontiguousarray(np.stack([a, b], 1)), out)
    with np.errstate(all="ignore"):
        ref = (a / b).astype(np.float32)
    ok_domain = (np.abs(a) == 0) | ((np.abs(a) >= 2.0 ** -100) & (np.abs(a) <= 2.0 ** 60))
    bad_div += int(((out.view(np.uint32) != ref.view(np.uint32)) & ok_domain).sum())
    x = np.ldexp(mant(), rng.randint(-96, 127, n)).astype(np.float32)
    if kind == 1:
        y = np.ldexp(mant(), rng.randint(-47, 63, n)).astype(np.float32)
        x = np.nextafter((y * y).astype(np.float32), np.float32(np.inf) * sgn()).astype(np.float32)
    dbg(7, np.ascontiguousarray(x), out)
    bad_sqrt += int((out.view(np.uint32) != np.sqrt(x).astype(np.float32).view(np.uint32)).sum())
    tested += n
    if r % 32 == 31:
        print("round %d: %.3g operands per operation, division mismatches %d, square-root mismatches %d (%.0f s)" % (r + 1, tested, bad_div, bad_sqrt, time.time() - t0), flush=True)
print("IN DOMAIN: %.3g divisions, %d mismatches; %.3g square roots, %d mismatches" % (tested, bad_div, tested, bad_sqrt))
# outside the domain: report only
a = (rng.randint(1, 1 << 23, n).astype(np.uint32)).view(np.float32)                   # denormal numerators
b = np.ldexp(mant(), rng.randint(-10, 11, n)).astype(np.float32)
out = np.empty(n, np.float32)
dbg(6, np.ascontiguousarray(np.stack([a, b], 1)), out)
print("OUTSIDE (denormal numerators / normal denominators): %d of %d differ from IEEE" % (int((out.view(np.uint32) != (a / b).astype(np.float32).view(np.uint32)).sum()), n))
a2 = np.ldexp(mant(), rng.randint(-126, -100, n)).astype(np.float32); b2 = np.ldexp(mant(), rng.randint(10, 60, n)).astype(np.float32)
dbg(6, np.ascontiguousarray(np.stack([a2, b2], 1)), out)
with np.errstate(all="ignore"):
    print("OUTSIDE (quotients below the normal range): %d of %d differ from IEEE" % (int((out.view(np.uint32) != (a2 / b2).astype(np.float32).view(np.uint32)).sum()), n))
x = (rng.randint(1, 1 << 23, n).astype(np.uint32)).view(np.float32)
dbg(7, np.ascontiguousarray(x), out)
print("OUTSIDE (square roots of denormals): %d of %d differ from IEEE" % (int((out.view(np.uint32) != np.sqrt(x).astype(np.float32).view(np.uint32)).sum()), n))
assert bad_div == 0 and bad_sqrt == 0
env.close()
