"""Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libcmad_hip.so")
SOURCES = ["cmad_hip.hip"]
HEADERS = ["cm_device.hpp", "cm_structured.hpp", os.path.join("..", "..", "include", "cmad_hip.h")]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), "-O3", "-std=c++20", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose:
        print(" ".join(cmd))
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
