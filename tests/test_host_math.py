"""CPU coverage of the hand-derived kernel arithmetic: cmad_amd/csrc/cm_device.hpp compiled for the
host (tests/native/host_harness.cpp) against the dual-number oracle.  The same scenarios run on the
GPU through the C-ABI in tests/test_gpu_update.py / test_gpu_sensitivities.py."""
import pytest

import oracle_lib as ol
import parity_cases as pc

BACKEND = pc.HostBackend()


@pytest.mark.parametrize("ls", [False, True])
@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS)
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_update(def_type, yield_kind, kw, rot, ls):
    pc.check_update(BACKEND, pc.Scenario(def_type, yield_kind, kw, rot, ls, B=768))


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_tangent(def_type, yield_kind, kw, rot):
    pc.check_tangent(BACKEND, pc.Scenario(def_type, yield_kind, kw, rot, False, B=256))


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("yield_kind,kw", pc.YIELDS[:3])
@pytest.mark.parametrize("def_type", [ol.FULL_3D, ol.PLANE_STRESS])
def test_vjp(def_type, yield_kind, kw, rot):
    pc.check_vjp(BACKEND, pc.Scenario(def_type, yield_kind, kw, rot, False, B=256))
