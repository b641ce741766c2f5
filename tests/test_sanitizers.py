"""AddressSanitizer + UBSan over the host build of the kernel arithmetic (GPU sanitizers are unavailable on the
pool; the per-point code is the same source as the HIP kernels, cmad_amd/csrc/cm_device.hpp + cm_structured.hpp).
Runs a small driver in a subprocess with the ASan runtime preloaded."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_math_under_asan_ubsan():
    import host_harness_lib as hh
    so = hh.build(sanitize=True)
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not found")
    driver = textwrap.dedent(f"""
        import sys, ctypes as C
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
        import numpy as np
        import host_harness_lib as hh
        hh._SO = {so!r}
        import oracle_lib as ol, parity_cases as pc
        from cmad_amd.models.device import build_desc, NewtonSettings
        from cmad_amd.synthetic import gauss_point_batch
        be = pc.HostBackend()
        for def_type in (ol.FULL_3D, ol.PLANE_STRESS):
            for yk, kw in pc.YIELDS[:3]:
                for ls in (False, True):
                    sc = pc.Scenario(def_type, yk, kw, True, ls, B=67)
                    be.update(sc, sc.gradu, sc.xi1, tangent=True)
                    be.vjp(sc, sc.gradu, sc.xi1, sc.xi2, np.ones((6, 67)))
        pc.check_edge_cases(be)
        pc.check_hybrid_nn(be, ol.FULL_3D, B=33)
        print("SANITIZED-OK")
    """)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    res = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0 and "SANITIZED-OK" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])
