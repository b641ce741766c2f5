"""The C-ABI is usable without Python or torch: examples/c_abi_demo.cpp links against libcmad_hip.so with only the
HIP runtime.  CPU: the header is valid C99 and the demo compiles and links; GPU: it runs and its own checks pass."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cmad_amd", "csrc")


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _build_demo(out):
    from cmad_amd import build
    build.build()                                          # no-op when the in-tree library is current
    cmd = [_hipcc(), "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_demo.cpp"), "-L", CSRC, "-lcmad_hip", f"-Wl,-rpath,{CSRC}", "-o", out]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return out


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "use_header.c"
    src.write_text('#include "cmad_hip.h"\nint main(void) { cm_model_desc m; (void)m; return cm_abi_version() > 0 ? 0 : 1; }\n')
    res = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                          "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_demo_compiles_and_links(tmp_path):
    exe = _build_demo(str(tmp_path / "c_abi_demo"))
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libcmad_hip.so" in needed and "torch" not in needed and "python" not in needed.lower()


@pytest.mark.gpu
def test_demo_runs_on_the_gpu(tmp_path):
    exe = _build_demo(str(tmp_path / "c_abi_demo"))
    res = subprocess.run([exe, "300000"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "C-ABI demo: OK" in res.stdout and "unconverged 0" in res.stdout
