"""Child process of tests/test_gpu_pool.py (the library reads its CM_DEBUG_POOL_* knobs once per process, so every
configuration of the dynamic chunk assignment gets a fresh interpreter).  TEST INFRASTRUCTURE: checks the work-pool route of
`cm_update` (`k_update_pool`, cmad_amd/csrc/cmad_hip.hip) -- the dynamic ticket counters in particular -- for completeness
(sentinel-prefilled outputs), against the static assignment (bitwise), under HIP-graph capture and with a replayed graph
overlapping eager launches on a second stream.

    python tests/pool_child.py <surface: hosford|hybrid> <case: complete|graph|streams> <B>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np          # noqa: E402
import torch                # noqa: E402

F64 = dict(dtype=torch.float64, device="cuda")


def evaluator(surface):
    """The two work-pool configurations: Hosford a = 100 on the reference's iteration from x_prev (CM_SOLVER_GENERAL_NEWTON:
    the default warm start needs no pool) and the hybrid Hill + network surface, BASELINE configs[2] / [3] settings."""
    from cmad_amd.models.device import DeviceEvaluator, HybridHillEffectiveStress, NewtonSettings, build_desc
    from cmad_amd.synthetic import al7079_hybrid_setup, hosford_values
    if surface == "hosford":
        values = hosford_values()
        nt = NewtonSettings.traced(max_iters=500, abs_tol=1e-12, rel_tol=1e-12, line_search_settings={"max evals": 100})
        nt.j2_radial_line = False
        return DeviceEvaluator(*build_desc(values, newton=nt)), (values, None), 2e-3, dict(max_iters=500, ls_max_evals=100)
    if surface == "barlat":                     # Yld2004-18p, Al7079 coefficients, a = 8 (screened route only: never on the pool)
        import parity_cases as pc
        import oracle_lib as ol
        values = ol.j2_voce_values(yield_kind="barlat", barlat=pc.AL7079_BARLAT[:18] + [8.0])
        nt = NewtonSettings.traced(max_iters=20, abs_tol=1e-12, rel_tol=1e-12)
        return DeviceEvaluator(*build_desc(values, newton=nt)), (values, None), 1e-3, dict(max_iters=20, ls_max_evals=4)
    icnn, values = al7079_hybrid_setup()
    nt = NewtonSettings.traced(max_iters=50, abs_tol=1e-12, rel_tol=1e-12, line_search_settings={"max evals": 10})
    return (DeviceEvaluator(*build_desc(values, newton=nt, hybrid=HybridHillEffectiveStress(icnn))), (values, icnn), 525.0 / 70.2e3,
            dict(max_iters=50, ls_max_evals=10))


def sentinel_outputs(B):
    return {"xi": torch.full((7, B), float("nan"), **F64), "sigma": torch.full((6, B), float("nan"), **F64),
            "status": torch.full((B,), -1, dtype=torch.int32, device="cuda")}


def run(ev, gradu, xi_prev, out=None):
    B = gradu.shape[1]
    out = out if out is not None else sentinel_outputs(B)
    ev.update(gradu, xi_prev, want_status=True, out=out)
    return out


def assert_complete(out, what):
    assert not bool(torch.isnan(out["xi"]).any()), f"{what}: unwritten state entries"
    assert not bool(torch.isnan(out["sigma"]).any()), f"{what}: unwritten stress entries"
    assert not bool((out["status"] == -1).any()), f"{what}: unwritten status words"
    conv = ((out["status"].to(torch.int64) >> 16) & 1).double().mean().item()
    assert conv > 0.999, f"{what}: converged fraction {conv}"


def same(a, b):
    return all(torch.equal(a[k], b[k]) for k in ("xi", "sigma", "status"))


def oracle_sample(ev, values, settings, g_host, out, n=384):
    import oracle_lib as ol
    B = g_host.shape[1]
    idx = np.sort(np.random.default_rng(5).choice(B, n, replace=False))
    st = ol.newton_settings(abs_tol=1e-12, rel_tol=1e-12, ls_kind=ol.LS_TRACED, **settings)
    mat = evaluator_material(values)
    xi_o, sig_o, it_o, cv_o = mat.update_batch(st, g_host[:, idx], np.zeros((7, n)))
    t = torch.from_numpy(idx).cuda()
    np.testing.assert_allclose(out["xi"][:, t].cpu().numpy(), xi_o, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(out["sigma"][:, t].cpu().numpy(), sig_o, rtol=1e-8, atol=1e-8 * np.abs(sig_o).max())
    same_its = ((out["status"][t].cpu().numpy().astype(np.uint32) & 0xFFFF) == it_o).mean()
    assert same_its > 0.97, same_its


def evaluator_material(values_icnn):
    import oracle_lib as ol
    values, icnn = values_icnn
    return ol.Material(values, nn=icnn.pack_for_device()) if icnn is not None else ol.Material(values)


def main():
    surface, case, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
    from cmad_amd.synthetic import gauss_point_batch
    ev, values, eps_y, settings = evaluator(surface)
    assert ev.pool_route(B) or ev.screened(B), "configuration is on neither the work-pool nor the screened route"
    g_host = gauss_point_batch(B, seed=77, eps_y=eps_y)
    gradu = torch.from_numpy(g_host).cuda()
    xi_prev = torch.zeros((7, B), **F64)

    if case == "complete":
        # every point written exactly as the static assignment writes it (slices below the dynamic threshold; the kernel's
        # results do not depend on which lane of which wavefront computes a point), and a sample agrees with the oracle
        out = run(ev, gradu, xi_prev)
        torch.cuda.synchronize()
        assert_complete(out, "dynamic assignment")
        again = run(ev, gradu, xi_prev)
        assert same(out, again), "two launches of the dynamic assignment differ"
        if os.environ.get("CM_DEBUG_POOL_DYNAMIC_MIN") is None:          # slices of a full-size batch run the static assignment
            for lo, n in ((0, 65_536), (B // 2 + 1, 65_537), (B - 40_001, 40_001)):
                part = run(ev, gradu[:, lo:lo + n].contiguous(), xi_prev[:, lo:lo + n].contiguous())
                assert_complete(part, "static assignment")
                assert all(torch.equal(part[k], out[k][..., lo:lo + n]) for k in ("xi", "sigma", "status")), (lo, n)
        oracle_sample(ev, values, settings, g_host, out)
        print("complete ok")
        return

    g2_host = gauss_point_batch(B, seed=78, eps_y=eps_y, dev_scale=5.0)
    gradu2 = torch.from_numpy(g2_host).cuda()
    ref1 = run(ev, gradu, xi_prev)                     # eager references, one launch at a time
    ref2 = run(ev, gradu2, xi_prev)
    torch.cuda.synchronize()
    assert_complete(ref1, "eager"); assert_complete(ref2, "eager")

    # capture cm_update on the pool route into a HIP graph (static input / output buffers), replay it on new inputs
    g_in = gradu.clone()
    g_out = sentinel_outputs(B)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run(ev, g_in, xi_prev, g_out)                  # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        run(ev, g_in, xi_prev, g_out)

    def reset(o):
        o["xi"].fill_(float("nan")); o["sigma"].fill_(float("nan")); o["status"].fill_(-1)

    if case == "graph":
        for src, ref in ((gradu2, ref2), (gradu, ref1), (gradu2, ref2)):
            g_in.copy_(src)
            reset(g_out)
            graph.replay()
            torch.cuda.synchronize()
            assert_complete(g_out, "graph replay")
            assert same(g_out, ref), "graph replay differs from the eager launch"
        oracle_sample(ev, values, settings, g2_host, g_out)
        print("graph ok")
        return

    if case == "streams":
        # the replayed graph (stream A, inputs 1) overlaps eager launches on stream B (inputs 2) and on the default stream
        # (inputs 2 again): every launch has a counter of its own, so each one still hands every chunk out exactly once
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        e_out, d_out = sentinel_outputs(B), sentinel_outputs(B)
        g_in.copy_(gradu)
        torch.cuda.synchronize()
        for rep in range(12):
            reset(g_out); reset(e_out); reset(d_out)
            torch.cuda.synchronize()
            with torch.cuda.stream(sa):
                graph.replay()
            with torch.cuda.stream(sb):
                run(ev, gradu2, xi_prev, e_out)
            run(ev, gradu2, xi_prev, d_out)
            with torch.cuda.stream(sa):
                graph.replay()                          # a second replay queued behind the first, still overlapping B
            torch.cuda.synchronize()
            for o, ref, what in ((g_out, ref1, "graph on stream A"), (e_out, ref2, "eager on stream B"), (d_out, ref2, "eager on the default stream")):
                assert_complete(o, what)
                assert same(o, ref), f"{what}: differs from the one-at-a-time result (repetition {rep})"
        print("streams ok")
        return
    raise SystemExit(f"unknown case {case}")


if __name__ == "__main__":
    main()
