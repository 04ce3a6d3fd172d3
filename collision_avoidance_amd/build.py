"""Builds collision_avoidance_amd/libcaenv.so (HIP, gfx950 only) in-tree with hipcc."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcaenv.so")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("ca_env.hip", "ca_kernels.h", "ca_common.h", "ca_lp.h", "ca_lines.h", "ca_nbr.h", "ca_step.h", "ca_quad.h", "ca_pair.h",
                                                    "ca_alan.h", "ca_obs.h", "ca_math.h")] + \
          [os.path.join(os.path.dirname(_HERE), "include", "ca_env.h")]
# -ffp-contract=off: no FMA contraction -- the numerics contract shared with the parity oracle.
# -fno-slp-vectorize: keeps the compiler from packing adjacent scalar fp32 adds / multiplies into v_pk_add_f32 /
#   v_pk_mul_f32.  On gfx950 a packed instruction issues every ~4.3 cycles against ~2.7 for v_add_f32 / v_mul_f32
#   (profiles/r02_valu_issue_rates_microbench.txt), so two scalar operations cost 5.4 cycles and the packed one 4.3 PLUS
#   the v_mov shuffles that line the operands up in register pairs (18 % of the observation kernel's pair loop): these
#   issue-bound kernels run 10 % faster without it (obs_kernel 80.5 -> 69.6 us, step_kernel 72.0 -> 68.9 us at C3; same
#   IEEE operations, bit-identical results; profiles/r03_e_no_slp_packing.txt).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-slp-vectorize", "-Wno-unused-value"]


def source_sha():
    """Hash of the kernel sources: profiles record it, and bench.py only quotes a counter profile whose hash is
    that of the sources the loaded library was built from."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode())   # the compiler flags are part of what a profile was taken from
    for f in sorted(SOURCES):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def loaded_sha():
    """The source hash compiled into the library that is actually loaded (ca_source_sha)."""
    import ctypes
    from . import _lib
    L = _lib.load()
    L.ca_source_sha.restype = ctypes.c_char_p
    return L.ca_source_sha().decode()


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libcaenv.so)")


def is_stale():
    """The library is missing, or was built from other sources / flags than the ones on disk: its compiled-in hash (CA_SRC_SHA,
    what ca_source_sha() returns) is looked for in the file itself -- time stamps say nothing after a checkout."""
    if not os.path.exists(LIB_PATH):
        return True
    with open(LIB_PATH, "rb") as fh:
        return source_sha().encode() not in fh.read()


def build(force=False, verbose=False):
    """Compile the HIP library if it is missing or older than its sources; returns its path."""
    if not force and not is_stale():
        return LIB_PATH
    # the hash of what is being compiled goes INTO the library (ca_source_sha()): a bench line or a counter profile
    # then names the sources of the code that ran, not of whatever lies on disk next to it
    cmd = [hipcc()] + HIPCC_FLAGS + ['-DCA_SRC_SHA="%s"' % source_sha(), "-o", LIB_PATH, SOURCES[0]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
