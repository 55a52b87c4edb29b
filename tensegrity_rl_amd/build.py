"""Build libtensegrity.so (HIP, gfx950) in-tree with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "tgr_sim.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "tgr_device.h"), os.path.join(os.path.dirname(HERE), "include", "tensegrity.h")]
LIB = os.path.join(HERE, "libtensegrity.so")


def lib_path():
    return LIB


def needs_build():
    return not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-o", LIB, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
