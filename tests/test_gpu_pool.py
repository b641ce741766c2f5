"""The work-pool route of `cm_update` (cmad_amd/csrc/cmad_hip.hip `k_update_pool`: the iteration-bound configurations -- the
network surfaces, Hosford on the reference's iteration under the line search) and its dynamic chunk assignment: completeness on
sentinel-prefilled outputs, bitwise agreement with the static assignment, HIP-graph capture / replay on new inputs, and a
replayed graph overlapping eager launches on other streams.  Each case runs tests/pool_child.py in a fresh interpreter because
the library reads its CM_DEBUG_POOL_* knobs once per process:
  CM_DEBUG_POOL_DYNAMIC_MIN=1   draw chunks from the device counter at any batch size (default: >= 2048 points per resident wavefront)
  CM_DEBUG_POOL_SLOTS=n         shrink the per-stream / per-capture counter tables to n entries (n = 1: the second stream and the
                                second captured launch fall back to the static assignment)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "pool_child.py")
pytestmark = pytest.mark.gpu


def child(surface, case, B, screened=False, **env):
    e = {k: v for k, v in os.environ.items() if not k.startswith("CM_DEBUG_")}
    e["CM_DEBUG_NO_SCREEN"] = "0" if screened else "1"     # DeviceEvaluator.update hands cm_update_ws a workspace: keep the work pool
    e.update({k: str(v) for k, v in env.items()})
    res = subprocess.run([sys.executable, CHILD, surface, case, str(B)], capture_output=True, text=True, env=e, timeout=900)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    assert f"{case} ok" in res.stdout


@pytest.mark.parametrize("surface,B", [("hosford", 5_000_001), ("hybrid", 4_200_000)])
def test_full_size_dynamic_assignment_writes_every_point(surface, B):
    """B large enough for the ticket path with the library's own threshold (>= 2048 points per resident wavefront); odd B for
    Hosford -> 4-byte LDS-DMA pieces, even for the network surface -> 16-byte pieces.  Outputs prefilled with NaN / 0xFFFFFFFF:
    no sentinel survives, two launches agree bit for bit, slices (static assignment) reproduce the batch, sample vs oracle."""
    child(surface, "complete", B)


@pytest.mark.parametrize("surface", ["hosford", "hybrid"])
@pytest.mark.parametrize("B", [5_000, 200_003])
def test_dynamic_assignment_at_small_batches(surface, B):
    child(surface, "complete", B, CM_DEBUG_POOL_DYNAMIC_MIN=1)


@pytest.mark.parametrize("surface,B", [("hosford", 4_400_000), ("hybrid", 4_200_000)])
def test_pool_route_under_graph_capture_full_size(surface, B):
    """cm_update on the pool route (ticket path) captured into a HIP graph, replayed on new inputs: equal to the eager
    launches bit for bit, sample against the oracle."""
    child(surface, "graph", B)


@pytest.mark.parametrize("surface", ["hosford", "hybrid"])
def test_pool_route_under_graph_capture_small(surface):
    child(surface, "graph", 150_000, CM_DEBUG_POOL_DYNAMIC_MIN=1)


@pytest.mark.parametrize("surface", ["hosford", "hybrid"])
@pytest.mark.parametrize("slots", [None, 1, 0])
def test_replayed_graph_overlapping_eager_launches_on_other_streams(surface, slots):
    """Grids small enough to be resident together (about 1200 wavefronts each): a replayed graph on one stream, eager launches
    on a second stream and on the default stream, repeated -- every launch equals its one-at-a-time result bit for bit.  With
    the counter tables shrunk to one entry (the forced collision of a shared table) and to none, the launches that get no
    counter of their own run the static assignment; nothing is ever shared."""
    env = {"CM_DEBUG_POOL_DYNAMIC_MIN": 1}
    if slots is not None:
        env["CM_DEBUG_POOL_SLOTS"] = slots
    child(surface, "streams", 150_000, **env)


# ---- the screened route (cm_update_ws: k_screen + k_update_listed; the network surfaces and Barlat) through the same child:
# completeness on sentinel-prefilled outputs and launch-size independence, graph capture (a private workspace per captured
# launch), streams (one workspace per stream)
@pytest.mark.parametrize("surface,B", [("hybrid", 4_200_000), ("barlat", 1_000_001)])
def test_screened_route_writes_every_point(surface, B):
    child(surface, "complete", B, screened=True)


@pytest.mark.parametrize("surface,B", [("hybrid", 600_000), ("barlat", 300_000)])
def test_screened_route_under_graph_capture(surface, B):
    child(surface, "graph", B, screened=True)


@pytest.mark.parametrize("surface", ["hybrid", "barlat"])
def test_screened_route_on_overlapping_streams(surface):
    child(surface, "streams", 150_000, screened=True)
