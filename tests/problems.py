"""Test problems shared by the facade tests: the reference's J2AnalyticalProblem parameter trees
(/root/reference/tests/support/test_problems.py:9-162) built with cmad_amd.parameters.Parameters."""
import copy

import numpy as np

from cmad_amd.parameters import Parameters
from cmad_amd.parameters.parameters import tree_map


def params_J2_voce(flat_param_values=(200e3, 0.3, 200., 200., 20.), scale_params=True, yield_kind="J2"):
    E, nu, Y, S, D = flat_param_values
    if yield_kind == "J2":
        eff = {"J2": 0.}
    elif yield_kind == "hill":
        eff = {"hill": dict(zip("FGHLMN", [0.5] * 6))}
    else:
        eff = {"hosford": {"a": 4.}}
    values = {
        "rotation matrix": np.eye(3),
        "elastic": {"E": E, "nu": nu},
        "plastic": {"effective stress": eff,
                    "flow stress": {"initial yield": {"Y": Y}, "hardening": {"voce": {"S": S, "D": D}}}}}
    flags = tree_map(lambda a: False, copy.deepcopy(values))
    flags["plastic"]["flow stress"] = tree_map(lambda x: True, flags["plastic"]["flow stress"])
    transforms = tree_map(lambda a: None, copy.deepcopy(values))
    if scale_params:
        fs = transforms["plastic"]["flow stress"]
        fs["initial yield"]["Y"] = np.array([200.])
        fs["hardening"]["voce"]["S"] = np.array([100., 300.])
        fs["hardening"]["voce"]["D"] = np.array([10., 30.])
    return Parameters(values, flags, transforms)


def plane_stress_F(strain_increment=0.02, num_pts_per_increment=50):
    """tests/objectives/test_J2_fd_checks.py:266-289 (0.02) / test_calibrations.py:60-80 (0.01)."""
    init = strain_increment / num_pts_per_increment
    eps_xx = np.r_[np.zeros(1), np.linspace(init, strain_increment, num_pts_per_increment),
                   np.ones(num_pts_per_increment) * strain_increment]
    eps_yy = np.r_[np.zeros(1), np.zeros(num_pts_per_increment), np.linspace(init, strain_increment, num_pts_per_increment)]
    n = 2 * num_pts_per_increment
    F = np.repeat(np.eye(2)[:, :, None], n + 1, axis=2)
    F[0, 0, :] += eps_xx[:n + 1]
    F[1, 1, :] += eps_yy[:n + 1]
    return F


def extended_leaf_problem(model_cls, yield_kind, active_rotation):
    """PLANE_STRESS calibration history whose active leaves lie outside the 12 native kernel parameters: the Hosford exponent
    and/or the rotation matrix (differentiated by forward-mode evaluation of the whole model, cm_param_blocks)."""
    from cmad_amd.models import DefType
    from cmad_amd.qois import Calibration
    base = params_J2_voce(yield_kind=yield_kind, scale_params=False)
    values = copy.deepcopy(base.values)
    th = 0.35
    values["rotation matrix"] = np.array([[np.cos(th), -np.sin(th), 0.], [np.sin(th), np.cos(th), 0.], [0., 0., 1.]])
    if yield_kind == "hill":
        values["plastic"]["effective stress"]["hill"] = dict(zip("FGHLMN", [0.35, 0.55, 0.6, 1.4, 1.6, 1.7]))
    flags = tree_map(lambda a: False, copy.deepcopy(values))
    flags["plastic"]["flow stress"]["initial yield"]["Y"] = True
    if yield_kind == "hosford":
        flags["plastic"]["effective stress"]["hosford"]["a"] = True
    if active_rotation:
        flags["rotation matrix"] = True                        # an array leaf is active as a whole: nine entries
    params = Parameters(values, flags, tree_map(lambda a: None, copy.deepcopy(values)))
    F = plane_stress_F(0.02, 3)
    model = model_cls(params, DefType.PLANE_STRESS)
    n = F.shape[2]
    t = np.linspace(0.0, 1.0, n)
    data = np.zeros((3, 3, n))
    data[0, 0] = 260.0 * np.tanh(4 * t); data[1, 1] = 180.0 * t; data[0, 1] = data[1, 0] = 15.0 * np.sin(3 * t)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.; weight[0, 1] = weight[1, 0] = 0.7
    return model, Calibration(model, data, weight), F
