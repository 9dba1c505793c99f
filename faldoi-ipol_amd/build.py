"""Builds libfaldoi_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the repo snapshot.  hipcc cross-compiles
gfx950 code objects without a GPU, so this runs in the CPU-only build container too.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "libfaldoi_hip.so")
SOURCES = [os.path.join(HERE, "csrc", "faldoi_hip.hip")]
DEPS = SOURCES + [os.path.join(HERE, "csrc", f) for f in ("faldoi_device.hpp", "tvl1_kernels.hpp")] + [
    os.path.join(ROOT, "include", "faldoi_hip.h")]

# -ffp-contract=off: parity with the reference CPU path needs un-fused multiply-adds.
# fp32 divide / sqrt stay on hipcc's default correctly-rounded expansions; denormals are kept.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-fno-gpu-flush-denormals-to-zero",
               # the SLP vectoriser packs fp32 pairs into v_pk_* ops, which on gfx950 cost two plain ops
               # plus the register moves that assemble the pairs (tools/ubench_valu.hip) -- a net loss here
               "-fno-slp-vectorize"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    cmd = [hipcc()] + HIPCC_FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"),
                                    "-o", SO] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
