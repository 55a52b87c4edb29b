"""Build libtensegrity.so (HIP, gfx950) in-tree with hipcc.

Staleness is decided by a hash of the sources kept next to the library (mtimes do not survive a copy to another box);
concurrent ranks serialise on a lock file and publish the library with an atomic rename."""
import fcntl
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "tgr_sim.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "tgr_device.h"), os.path.join(HERE, "csrc", "tgr_row.h"), os.path.join(HERE, "csrc", "tgr_spmd.h"),
        os.path.join(os.path.dirname(HERE), "include", "tensegrity.h")]
LIB = os.path.join(HERE, "libtensegrity.so")
STAMP = LIB + ".srchash"
# -fno-slp-vectorize: the SLP vectoriser pairs the row kernels' fp32 arithmetic into v_pk_* instructions, which take as long as two scalar
# ones on this hardware would if the pairs did not have to be assembled first -- 690 of the 4400 instructions of the Newton loop were
# register moves doing that, and 135 VGPRs spilled (37 without it); measured +7 % env-steps/s
# -amdgpu-sched-strategy=max-ilp: a wavefront of the row kernels is alone with one other on its SIMD and issues VALU 61 % of its cycles -- what is
# left are dependent chains and LDS round trips, and the ILP-first scheduling strategy interleaves them better than the default
# (occupancy-first) one: live-row launch -2.3 %, no spills (profiles/round4_ab_scheduling.log)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def lib_path():
    return LIB


def _src_hash():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    try:
        return open(STAMP).read().strip() != _src_hash()
    except OSError:
        return True


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # another rank built it while we waited
                return LIB
            tmp = f"{LIB}.tmp.{os.getpid()}"
            cmd = [hipcc] + FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if verbose else []) + ["-o", tmp, SRC]
            stamp = _src_hash()  # of the sources as they are NOW: an edit made while the compiler runs must leave the library stale
            subprocess.check_call(cmd)
            if _src_hash() != stamp:
                stamp = "stale: sources changed during the build"
            os.replace(tmp, LIB)
            with open(STAMP + ".tmp", "w") as f:
                f.write(stamp)
            os.replace(STAMP + ".tmp", STAMP)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
