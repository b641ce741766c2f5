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
        # the resumable Newton of the work-pool kernels (cm_pool.hpp)
        hh.set_passes(True)
        for yk, kw in pc.YIELDS[:3]:
            sc = pc.Scenario(ol.PLANE_STRESS, yk, kw, True, True, B=35)
            be.update(sc, sc.gradu, sc.xi1)
        hh.set_passes(False)
        # whole-history kernels' per-point code, extended parameter blocks, second derivatives, the 12-dof rate form
        from host_facade import HostHistoryEngine
        mk_desc = lambda values, dt, mk: build_desc(values, def_type=dt, model_kind=mk)
        for rate in (False, True):
            pc.check_history_second_order(lambda desc, info: HostHistoryEngine(desc=desc, info=info), mk_desc,
                                          ol.PLANE_STRESS, "hill", pc.YIELDS[1][1], rate=rate)
            pc.check_param_blocks(hh.param_blocks, ol.UNIAXIAL_STRESS, "hosford", pc.YIELDS[2][1], rate=rate)
        pc.check_param_blocks_network(hh.param_blocks, ol.FULL_3D, scaled=True)
        pc.check_second_derivs_network(hh.hessians, ol.PLANE_STRESS, scaled=False)
        pc.check_barlat_generic(hh.hessians, hh.param_blocks, ol.UNIAXIAL_STRESS)
        pc.check_rate_uniaxial_dual(hh.hessians, "hill", pc.YIELDS[1][1], 1, True)
        pc.check_rate_model(lambda desc, info, g, gp, xp: hh.update_rate(desc, g, gp, xp, 12), ol.UNIAXIAL_STRESS, "J2", dict(), True, True, B=33)
        print("SANITIZED-OK")
    """)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    res = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, env=env, timeout=1500)
    assert res.returncode == 0 and "SANITIZED-OK" in res.stdout, (res.stdout[-2000:], res.stderr[-4000:])
