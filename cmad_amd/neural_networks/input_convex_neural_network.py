"""Input-convex neural network used as the yield-surface discrepancy of the hybrid Hill + NN effective
stress.  Host mirror of /root/reference/cmad/neural_networks/input_convex_neural_network.py:13-109:
`AffineScaler`, seeded parameter initialisation (legacy `np.random.seed`, :85-109), and the forward
functions restated in numpy (`forward` :58-69, `input_symmetric_forward_with_offset` :36-48) for host-side
use.  `pack_for_device()` flattens weights + scalers into the layout the HIP kernel (and the oracle) read."""
from __future__ import annotations

import numpy as np


class AffineScaler:
    """Per-feature affine scaler onto a target range (reference :13-33)."""

    def __init__(self, feature_range=(-1.0, 1.0)):
        self.feature_range = feature_range

    def fit(self, samples):
        low, high = self.feature_range
        samples = np.asarray(samples, dtype=float)
        data_min = samples.min(axis=0)
        data_range = samples.max(axis=0) - data_min
        data_range[data_range == 0.0] = 1.0
        self.scale_ = (high - low) / data_range
        self.min_ = low - data_min * self.scale_
        return self


def softplus(x):
    return np.logaddexp(x, 0.0)


def forward(x, params):
    *x_hidden, x_last = params["x params"]
    *z_hidden, z_last = params["z params"]
    z = softplus(x @ x_hidden[0]["weights"] + x_hidden[0]["biases"])
    for x_layer, z_layer in zip(x_hidden[1:], z_hidden):
        z = softplus(z @ z_layer["weights"] + x @ x_layer["weights"] + x_layer["biases"])
    return z @ z_last["weights"] + x @ x_last["weights"] + x_last["biases"]


def input_symmetric_forward(x, params):
    zero = forward(np.zeros_like(x), params)
    return 0.5 * ((forward(x, params) - zero) + (forward(-x, params) - zero))


def input_symmetric_forward_with_offset(x, params, input_scaler, output_scaler):
    xs = input_scaler.scale_ * x + input_scaler.min_
    return (input_symmetric_forward(xs, params) - output_scaler.min_) / output_scaler.scale_


class InputConvexNeuralNetwork:
    def __init__(self, layer_widths, input_scaler, output_scaler, seed: int = 22):
        self.layer_widths = list(layer_widths)
        self.input_scaler, self.output_scaler = input_scaler, output_scaler
        self._init_params(layer_widths, seed)

    def evaluate(self, x, params):
        return input_symmetric_forward_with_offset(x, params, self.input_scaler, self.output_scaler)

    def _init_params(self, layer_widths, seed: int):
        np.random.seed(seed)
        nx, nz = len(layer_widths) - 1, len(layer_widths) - 2
        x_params, z_params = [None] * nx, [None] * nz
        for idx, num_out in enumerate(layer_widths[1:]):
            num_in = layer_widths[0]
            x_params[idx] = dict(weights=np.random.normal(size=(num_in, num_out)) * np.sqrt(2. / num_in),
                                 biases=np.ones(num_out))
        for idx, (num_in, num_out) in enumerate(zip(layer_widths[1:-1], layer_widths[2:])):
            z_params[idx] = dict(weights=np.abs(np.random.normal(size=(num_in, num_out)) * np.sqrt(2. / num_in)))
        self.x_params, self.z_params = x_params, z_params

    @property
    def params(self):
        return {"x params": self.x_params, "z params": self.z_params}

    def pack_for_device(self, params=None):
        """(layer_widths, flat float64 array): x-layers (W[in][out] row-major, b[out]) in order, z-layers
        (W row-major) in order, then in_scale[nin], in_min[nin], out_scale, out_min."""
        params = params or self.params
        parts = []
        for layer in params["x params"]:
            parts += [np.asarray(layer["weights"], dtype=np.float64).ravel(), np.asarray(layer["biases"], dtype=np.float64).ravel()]
        for layer in params["z params"]:
            parts.append(np.asarray(layer["weights"], dtype=np.float64).ravel())
        nin = self.layer_widths[0]
        parts += [np.broadcast_to(np.asarray(self.input_scaler.scale_, dtype=np.float64), (nin,)).ravel(),
                  np.broadcast_to(np.asarray(self.input_scaler.min_, dtype=np.float64), (nin,)).ravel(),
                  np.atleast_1d(np.asarray(self.output_scaler.scale_, dtype=np.float64)).ravel()[:1],
                  np.atleast_1d(np.asarray(self.output_scaler.min_, dtype=np.float64)).ravel()[:1]]
        return self.layer_widths, np.concatenate(parts)
