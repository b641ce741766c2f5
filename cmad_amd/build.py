"""Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import tempfile
import time

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libcmad_hip.so")
SOURCES = ["cmad_hip.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))) + [os.path.join("..", "..", "include", "cmad_hip.h")]


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


NPARTS = 12         # cmad_hip.hip compiles in independent pieces selected by -DCM_PART=k (see its header comment)


def build(force=False, verbose=False, jobs=None):
    """hipcc --offload-arch=gfx950: the parts of cmad_hip.hip are compiled concurrently, then linked."""
    if not force and not is_stale():
        return LIB
    hipcc = hipcc_path()
    src = os.path.join(CSRC, SOURCES[0])
    # two builds of every part (see the top of cmad_hip.hip): BASE (no network hardening law, one-hidden-layer networks) and EXT
    # (network hardening law on J2 / Hill / Hosford; the plain hybrid surface with multi-layer networks; no rate form x dense
    # surface); the heavy base parts first so that the shorter EXT parts fill the tail of the schedule
    objs = [os.path.join(CSRC, f"cmad_hip_part{k}.o") for k in range(NPARTS)] + \
           [os.path.join(CSRC, f"cmad_hip_hnn_part{k}.o") for k in range(NPARTS)]
    cmds = [[hipcc, "-O3", "-std=c++20", "--offload-arch=gfx950", "-fPIC", f"-DCM_PART={k}", "-c", src, "-o", objs[k]]
            for k in range(NPARTS)] + \
           [[hipcc, "-O3", "-std=c++20", "--offload-arch=gfx950", "-fPIC", f"-DCM_PART={k}", "-DCM_HNN_VARIANT=1", "-DCM_RATE_DENSE=0",
             "-DCM_RATE_UNIAXIAL_DENSE=0", "-c", src, "-o", objs[NPARTS + k]] for k in range(NPARTS)]
    jobs = jobs or min(NPARTS, os.cpu_count() or 1)
    # longest compiles first (measured: the arithmetic-T parts 9 - 11 and the reverse-sweep parts 5 / 6 of either build), every
    # free slot refilled as soon as ANY running compile ends
    heavy = [NPARTS + 11, NPARTS + 10, 11, 10, NPARTS + 6, 5, NPARTS + 5, 9, NPARTS + 9, 2, 1]
    order = heavy + [k for k in range(2 * NPARTS) if k not in heavy]
    pending = [cmds[k] for k in order]
    procs, failed = [], []
    logs = {}
    while pending or procs:
        while pending and len(procs) < jobs:
            c = pending.pop(0)
            log = tempfile.TemporaryFile(mode="w+")
            procs.append((c, subprocess.Popen(c, stdout=log, stderr=subprocess.STDOUT, text=True), log))
        done = [t for t in procs if t[1].poll() is not None]
        if not done:
            time.sleep(0.2)
            continue
        for c, p, log in done:
            procs.remove((c, p, log))
            if p.returncode != 0:
                log.seek(0)
                failed.append(" ".join(c) + "\n" + log.read())
            log.close()
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    if verbose:
        print("\n".join(" ".join(c) for c in cmds + [link]))
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
