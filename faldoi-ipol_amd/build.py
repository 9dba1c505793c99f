"""Builds libfaldoi_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the repo snapshot.  hipcc cross-compiles
gfx950 code objects without a GPU, so this runs in the CPU-only build container too.
"""
import contextlib
import fcntl
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "libfaldoi_hip.so")
SOURCES = [os.path.join(HERE, "csrc", "faldoi_hip.hip")]
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))
              if f.endswith((".hip", ".hpp", ".inc"))) + [os.path.join(ROOT, "include", "faldoi_hip.h")]
BIN = os.path.join(HERE, "bin")
HOST = os.path.join(HERE, "host")
HOST_COMMON = [os.path.join(HOST, "image_io.cpp")]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HOST_PROGRAMS = {          # name -> (sources, needs libfaldoi_hip, extra compiler / linker arguments)
    "global_faldoi": ([os.path.join(HOST, "global_faldoi.cpp")], True, []),
    "local_faldoi": ([os.path.join(HOST, "local_faldoi.cpp")], True, []),
    "faldoi_flow": ([os.path.join(HOST, "faldoi_flow.cpp")], True, []),
    "sparse_flow": ([os.path.join(HOST, "sparse_flow.cpp")], False, []),
    "faldoi_imginfo": ([os.path.join(HOST, "imginfo.cpp")], False, []),
    # the multi-GPU batch host: one process per GPU, RCCL directly (plain g++ against the HIP runtime API and RCCL) ...
    "faldoi_batch": ([os.path.join(HOST, "faldoi_batch.cpp")], True,
                     ["-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROCM, "include"), "-L" + os.path.join(ROCM, "lib"),
                      "-lrccl", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-pthread"]),
    # ... and its rank logic alone for the CPU tests: pipes instead of RCCL, `flow + 1` instead of the solver
    "faldoi_batch_stub": ([os.path.join(HOST, "faldoi_batch.cpp")], False, ["-DFALDOI_BATCH_STUB", "-pthread"]),
}

# -ffp-contract=off: parity with the reference CPU path needs un-fused multiply-adds.
# fp32 divide / sqrt stay on hipcc's default correctly-rounded expansions; denormals are kept.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-fno-gpu-flush-denormals-to-zero",
               # the SLP vectoriser packs fp32 pairs into v_pk_* ops, which on gfx950 cost two plain ops
               # plus the register moves that assemble the pairs (tools/ubench_valu.hip) -- a net loss here
               "-fno-slp-vectorize",
               # keep the per-thread state arrays as individual registers: promoted to vector tuples,
               # every conditionally updated row costs a copy of the whole tuple set (180 v_mov_b64)
               "-mllvm", "-disable-promote-alloca-to-vector"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


SO_CHECK = os.path.join(HERE, "libfaldoi_hip_check.so")       # the same sources with -DFALDOI_NL_CHECK=1 (nltv_fused_kernel.hpp: nf_check)


def build_check(force=False):
    """The range-checked build of the library: every LDS index and slab offset of the fused NLTV kernel is compared with its extent
    (tests/test_gpu_round4.py runs the geometry sweep on it once, in a child process).  Not a product artefact; same lock, same
    freshness rule as the product library."""
    fresh = os.path.exists(SO_CHECK) and all(os.path.getmtime(d) <= os.path.getmtime(SO_CHECK) for d in DEPS)
    if fresh and not force:
        return SO_CHECK
    with _build_lock():
        tmp = SO_CHECK + ".tmp.%d" % os.getpid()
        cmd = [hipcc()] + HIPCC_FLAGS + ["-DFALDOI_NL_CHECK=1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"),
                                         "-o", tmp] + SOURCES
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on the range-checked build:\n" + r.stderr[-4000:])
        os.replace(tmp, SO_CHECK)
    return SO_CHECK


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in DEPS)


@contextlib.contextmanager
def _build_lock():
    """Serialises builds across processes (every rank of a multi-GPU launch imports this package);
    the output is written to a temporary name and renamed, so a reader never sees a partial file."""
    fd = os.open(os.path.join(HERE, ".build.lock"), os.O_CREAT | os.O_RDWR, 0o644)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        yield
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)


def build(force=False, verbose=False):
    extra = os.environ.get("FALDOI_EXTRA_FLAGS", "").split()      # development aid (bisecting kernels)
    if not force and not needs_build():
        return SO
    with _build_lock():
        if not force and not needs_build():          # another process built it while we waited
            return SO
        tmp = SO + ".tmp.%d" % os.getpid()
        cmd = [hipcc()] + HIPCC_FLAGS + extra + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"),
                                                "-o", tmp] + SOURCES
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, SO)
    return SO


def build_host(force=False, verbose=False):
    """The reference's command-line surface (C++17, g++): global_faldoi over the C ABI, sparse_flow."""
    os.makedirs(BIN, exist_ok=True)
    cxx = shutil.which("g++") or "g++"
    hdrs = [os.path.join(HOST, f) for f in ("image_io.hpp", "options.hpp")] + [os.path.join(ROOT, "include", "faldoi_hip.h")]
    out = {}
    with _build_lock():
        for name, (srcs, needs_lib, extra) in HOST_PROGRAMS.items():
            exe = os.path.join(BIN, name)
            out[name] = exe
            deps = srcs + HOST_COMMON + hdrs + ([SO] if needs_lib else [])
            if not force and os.path.exists(exe) and all(os.path.getmtime(d) <= os.path.getmtime(exe) for d in deps):
                continue
            tmp = exe + ".tmp.%d" % os.getpid()
            cmd = [cxx, "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + HOST, "-o", tmp] + srcs + HOST_COMMON
            if needs_lib:
                cmd += ["-L" + HERE, "-lfaldoi_hip", "-Wl,-rpath,$ORIGIN/.."]
            cmd += extra + ["-lz"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, exe)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
