"""Build recipe for libgpuq.so (gfx950 only) -- plain hipcc, no cmake, no torch extension.

`python arrow-ballista_amd/build.py` or `__graft_entry__.build()`.  hipcc cross-compiles gfx950
without a GPU.  The build fails when any interpreter kernel uses scratch memory: the per-lane
register file must stay in VGPRs (see csrc/gpuq_dev.h).
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libgpuq.so")
OBJ = os.path.join(HERE, "build")

HIP_SOURCES = ["kernels_scan.hip", "kernels_hash.hip", "kernels_sort.hip", "kernels_gen.hip"]
CPP_SOURCES = ["expr_compile.cpp", "capi.cpp"]
HEADERS = ["gpuq_dev.h", "gpuq_kernels.h", "expr_compile.h", "json.h", os.path.join("..", "..", "include", "gpuq.h")]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src in HIP_SOURCES + CPP_SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        objs.append(op)
        if not force and _newer(op, [sp] + hdrs):
            continue
        cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "-c", sp, "-o", op]
        if src.endswith(".hip"):
            cmd += ["--offload-arch=" + ARCH, "-Rpass-analysis=kernel-resource-usage"]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("hipcc failed on " + src)
        if src.endswith(".hip"):
            _check_scratch(src, r.stderr)
    if force or not _newer(OUT, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    return OUT


def _check_scratch(src, remarks):
    name = None
    bad = []
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        # the 16-slot instantiations (more than 8 input columns) may spill a few dwords; everything else must be clean
        if m and int(m.group(1)) != 0 and not ("ILi16E" in (name or "") and int(m.group(1)) <= 128):
            bad.append((name, int(m.group(1))))
    if bad:
        raise RuntimeError("%s: kernels use scratch memory (register file demoted): %s" % (src, bad))


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
