#!/usr/bin/env python3
"""Registers, scratch and code size of every kernel of libcaenv.so, read from the compiler's own metadata.

  python tools/kernel_resources.py [--check] [--filter SUBSTR] [extra hipcc flags ...]

Compiles collision_avoidance_amd/csrc/ca_env.hip for gfx950 with the product flags plus -save-temps into build/isa/
(hipcc cross-compiles without a GPU), parses the `amdhsa.kernels` metadata of the device assembly and prints one row
per kernel: VGPRs, SGPRs, scratch bytes per lane (`.private_segment_fixed_size`), spilled VGPRs / SGPRs, static LDS,
code bytes, vector-instruction count.  --check exits non-zero if a step / quad / observation kernel uses scratch: a
spill at the 128-VGPR limit turns into HBM traffic (round 1: 19 MB per launch) and must not come back silently
(tests/test_host_cpu.py runs this check when the assembly is already there).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from collision_avoidance_amd import build as b  # noqa: E402

OUT = os.path.join(ROOT, "build", "isa")
ASM = os.path.join(OUT, "ca_env-hip-amdgcn-amd-amdhsa-gfx950.s")
HOT = ("step_kernel", "quad_kernel", "pair_kernel", "obs_kernel", "nbr_kernel")   # kernels that must not spill


def compile_asm(extra=()):
    os.makedirs(OUT, exist_ok=True)
    cmd = [b.hipcc()] + b.HIPCC_FLAGS + list(extra) + ["-save-temps", "-o", os.path.join(OUT, "libcaenv_isa.so"), b.SOURCES[0]]
    subprocess.check_call(cmd, cwd=OUT)
    with open(os.path.join(OUT, "src_sha"), "w") as f:
        f.write(b.source_sha())


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True, check=True).stdout
        return out.splitlines()
    except Exception:
        return list(names)


def parse(asm_path=ASM):
    """[{name, vgpr, sgpr, scratch, vgpr_spill, sgpr_spill, lds, code_bytes, valu}] from the assembly file."""
    text = open(asm_path).read()
    # per function: "<sym>:" ... "; codeLenInByte = N" (the compiler's trailer); vector instructions counted in between
    valu, size = {}, {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?); codeLenInByte = (\d+)", text, re.S | re.M):
        valu[m.group(1)] = len(re.findall(r"^\s+v_", m.group(2), re.M))
        size[m.group(1)] = int(m.group(3))
    rows = []
    meta = text[text.rindex("amdhsa.kernels:"):]
    for blk in re.split(r"\n  - ", meta)[1:]:
        def g(key, default=0):
            mm = re.search(r"^    \.%s:\s+(\S+)" % key, blk, re.M)
            return mm.group(1) if mm else default
        sym = g("name", "?")
        rows.append(dict(sym=sym, vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")),
                         scratch=int(g("private_segment_fixed_size")), vgpr_spill=int(g("vgpr_spill_count")),
                         sgpr_spill=int(g("sgpr_spill_count")), lds=int(g("group_segment_fixed_size")),
                         code_bytes=size.get(sym, 0), valu=valu.get(sym, 0)))
    for r, n in zip(rows, demangle([r["sym"] for r in rows])):
        r["name"] = n.replace("void ca::", "").replace("(ca::StepArgs)", "").replace("(ca::ObsArgs)", "")
    return rows


def by_design(name):
    """The LDS-line-table variant step_kernel<K, BS, 0, *> keeps LP3's projected lines in a private array (ca_lp.h lp3:
    only the few lanes whose LP2 is infeasible touch it); every other hot kernel must run without scratch memory."""
    # (Rounds 3-4 also exempted the K = 10 register-line kernels of two and more waves with obstacle lists of 16 -- two of their
    # 14 line slots lived in scratch -- and quad_kernel<10, 512, 16>.  Round 5: the register allocator took the two rare stages'
    # `while (true)` loops for hot and the fully unrolled LP2 for cold; with the loops marked unlikely and the rare stages' constants
    # and addresses formed where they are used, EVERY instantiation that pick_variant can select for K <= 10 has ScratchSize 0.)
    return re.match(r"step_kernel<\d+, \d+, 0, (true|false)(, \w+)*>", name) is not None


def spilling(rows):
    return [r for r in rows if any(h in r["name"] for h in HOT) and not by_design(r["name"]) and (r["scratch"] or r["vgpr_spill"])]


def ensure_asm(extra=()):
    fresh = os.path.exists(ASM) and os.path.exists(os.path.join(OUT, "src_sha")) and \
        open(os.path.join(OUT, "src_sha")).read() == b.source_sha() and not extra
    if not fresh:
        compile_asm(extra)


def main():
    args = sys.argv[1:]
    check = "--check" in args
    flt = None
    if "--filter" in args:
        flt = args[args.index("--filter") + 1]
    extra = [a for a in args if a.startswith("-") and a not in ("--check", "--filter")]
    ensure_asm(extra)
    rows = parse()
    print("%-52s %5s %5s %8s %7s %7s %7s %9s %7s" % ("kernel", "VGPR", "SGPR", "scratch", "v-spill", "s-spill", "LDS", "code B", "VALU"))
    for r in sorted(rows, key=lambda r: r["name"]):
        if flt and flt not in r["name"]:
            continue
        print("%-52s %5d %5d %8d %7d %7d %7d %9d %7d" % (r["name"][:52], r["vgpr"], r["sgpr"], r["scratch"], r["vgpr_spill"],
                                                       r["sgpr_spill"], r["lds"], r["code_bytes"], r["valu"]))
    bad = spilling(rows)
    if bad:
        print("\nkernels of the hot path that use scratch memory:")
        for r in bad:
            print("   %s: %d B per lane, %d VGPRs spilled" % (r["name"], r["scratch"], r["vgpr_spill"]))
        if check:
            raise SystemExit(1)
    elif check:
        print("\nno hot-path kernel uses scratch memory")


if __name__ == "__main__":
    main()
