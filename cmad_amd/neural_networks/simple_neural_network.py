"""`SimpleNeuralNetwork`: a small fully connected network with sigmoid hidden units, used by the reference as a
hardening law (`hardening_funs={"neural network": nn.evaluate}`; /root/reference/cmad/neural_networks/
simple_neural_network.py:13-46, examples/noisy_calibration.py:245-252).  Host (numpy) mirror of its constructor,
parameter layout and initialisation; the kernels evaluate widths [1, H, 1] (`cm_model_desc.hnn_width`).

    evaluate(x, params) = output_scale * (forward(input_scale * x) - forward(0))
    forward: hidden layers sigmoid(x @ W + b), last layer x @ W + b
"""
from __future__ import annotations

from functools import partial

import numpy as np


def sigmoid(a):
    a = np.asarray(a, dtype=np.float64)
    e = np.exp(-np.abs(a))
    return np.where(a >= 0, 1.0 / (1.0 + e), e / (1.0 + e))


def forward(x, params):
    *hidden, last = params
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    for layer in hidden:
        x = sigmoid(x @ np.asarray(layer["weights"]) + np.asarray(layer["biases"]))
    return x @ np.asarray(last["weights"]) + np.asarray(last["biases"])


def forward_with_offset(x, params, input_scale, output_scale):
    xs = input_scale * np.atleast_1d(np.asarray(x, dtype=np.float64))
    return output_scale * (forward(xs, params) - forward(np.zeros_like(xs), params))


class SimpleNeuralNetwork:
    """`SimpleNeuralNetwork(layer_widths, input_scale, output_scale)`; `.params` is the list of per-layer
    {"weights", "biases"} dicts that goes under params["plastic"]["flow stress"]["hardening"]["neural network"],
    `.evaluate` the callable handed to the model as hardening_funs["neural network"]."""

    def __init__(self, layer_widths, input_scale: float = 1., output_scale: float = 1.):
        self.layer_widths = [int(w) for w in layer_widths]
        self.input_scale, self.output_scale = float(input_scale), float(output_scale)
        self._init_params(self.layer_widths)
        self.evaluate = partial(forward_with_offset, input_scale=self.input_scale, output_scale=self.output_scale)

    def _init_params(self, layer_widths, seed: int = 22):
        # abs initialisation for monotonic networks, biases one (reference :34-46)
        rng_state = np.random.get_state()
        np.random.seed(seed)
        self.params = [dict(weights=np.abs(np.random.normal(size=(n_in, n_out)) * np.sqrt(2. / n_in)), biases=np.ones(n_out))
                       for n_in, n_out in zip(layer_widths[:-1], layer_widths[1:])]
        np.random.set_state(rng_state)


def hardening_network_scales(hardening_funs):
    """(input_scale, output_scale) of the network behind hardening_funs["neural network"] -- a `SimpleNeuralNetwork.evaluate`
    (functools.partial of `forward_with_offset`) -- or None when the table has no such entry.  The table is the reference's
    lookup `combined_hardening_fun` indexes by the keys of params[...]["hardening"] (cmad/models/hardening.py:22-34): its
    "voce" / "linear" entries must be the built-in laws (`cmad_amd.models.hardening.get_hardening_funs()`), which the kernels
    evaluate themselves; any other callable has no kernel."""
    from ..models.hardening import linear_hardening, voce_hardening
    builtin = {"voce": voce_hardening, "linear": linear_hardening}
    for name, fun in hardening_funs.items():
        if name == "neural network":
            continue
        if name not in builtin or fun is not builtin[name]:
            raise NotImplementedError(f"hardening_funs[{name!r}]: only the built-in Voce / linear laws (cmad_amd.models.hardening."
                                      "get_hardening_funs()) and {'neural network': SimpleNeuralNetwork(...).evaluate} select a "
                                      "kernel; arbitrary callables cannot be traced into a HIP kernel")
    if "neural network" not in hardening_funs:
        return None
    fun = hardening_funs["neural network"]
    kw = getattr(fun, "keywords", None)
    if getattr(fun, "func", None) is not forward_with_offset or not kw or set(kw) != {"input_scale", "output_scale"}:
        raise NotImplementedError("hardening_funs['neural network'] must be SimpleNeuralNetwork(...).evaluate: arbitrary "
                                  "callables cannot be traced into a HIP kernel")
    return float(kw["input_scale"]), float(kw["output_scale"])


def pack_hardening_network(nn_params, input_scale, output_scale):
    """Device layout of a [1, H, 1] hardening network (include/cmad_hip.h, hnn_width): W1[H], b1[H], W2[H], b2, in_scale,
    out_scale, then sigmoid(b1[u]) (what forward(0) needs, constant over the points).  Returns (H, packed)."""
    if len(nn_params) != 2:
        raise ValueError("pack_hardening_network: one hidden layer; pack_hardening_network_deep handles [1, H1, ..., Hn, 1]")
    W1 = np.asarray(nn_params[0]["weights"], dtype=np.float64)
    b1 = np.asarray(nn_params[0]["biases"], dtype=np.float64).ravel()
    W2 = np.asarray(nn_params[1]["weights"], dtype=np.float64)
    b2 = np.asarray(nn_params[1]["biases"], dtype=np.float64).ravel()
    H = b1.size
    if W1.shape != (1, H) or W2.shape != (H, 1) or b2.size != 1:
        raise NotImplementedError(f"hardening network layer shapes {W1.shape}, {W2.shape}: expected (1, H) and (H, 1)")
    return H, np.concatenate([W1.ravel(), b1, W2.ravel(), b2, [input_scale, output_scale], sigmoid(b1)])


MAX_HIDDEN_LAYERS, MAX_HIDDEN_UNITS = 4, 64        # kHnnMaxHidden / kHnnMaxUnits of cmad_amd/csrc/cm_device.hpp


def pack_hardening_network_deep(nn_params, input_scale, output_scale):
    """Device layout of a hardening network with several hidden layers, widths [1, H1, ..., Hn, 1] (include/cmad_hip.h,
    hnn_nhidden >= 2): for every layer W[n_in][n_out] (row-major, as stored) and b[n_out]; then in_scale, out_scale and
    forward(0), the constant `forward_with_offset` subtracts.  Returns ([H1, ..., Hn], packed)."""
    widths = [int(np.asarray(layer["biases"]).size) for layer in nn_params[:-1]]
    if not (2 <= len(widths) <= MAX_HIDDEN_LAYERS) or sum(widths) > MAX_HIDDEN_UNITS:
        raise NotImplementedError(f"hardening network with hidden widths {widths}: the kernels evaluate 1 to {MAX_HIDDEN_LAYERS} "
                                  f"hidden layers with at most {MAX_HIDDEN_UNITS} hidden units in all")
    n_in, parts = 1, []
    for layer, n_out in zip(nn_params, widths + [1]):
        W = np.asarray(layer["weights"], dtype=np.float64)
        b = np.asarray(layer["biases"], dtype=np.float64).ravel()
        if W.shape != (n_in, n_out) or b.size != n_out:
            raise NotImplementedError(f"hardening network layer shapes {W.shape} / {b.shape}: expected ({n_in}, {n_out}) / ({n_out},)")
        parts += [W.ravel(), b]
        n_in = n_out
    f0 = float(np.asarray(forward(np.zeros(1), nn_params)).ravel()[0])
    return widths, np.concatenate(parts + [[input_scale, output_scale, f0]])
