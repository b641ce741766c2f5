"""`Parameters`: constitutive-model parameters held as a nested dict ("pytree") with per-leaf
active flags and transforms.  Host-side mirror of the reference's ``cmad.parameters.parameters``
(/root/reference/cmad/parameters/parameters.py:176-401) with the same method names, argument meaning
and in-place behaviour, re-implemented without JAX: the pytree utilities the reference takes from
``jax.tree_util`` / ``jax.flatten_util.ravel_pytree`` are restated here in ~40 lines of plain Python.

Flat ordering is JAX's dict-pytree ordering: keys sorted at every level, array leaves raveled in C
order (reference :214-227).  For the J2 test tree this gives
``[E, nu, J2, D, S, Y, R00..R22]`` (tests/support/test_problems.py:28-40 in the reference).
"""
from __future__ import annotations

from functools import partial
from typing import Any, Callable

import numpy as np

# --------------------------------------------------------------------------- pytree helpers


def _is_leaf_default(x) -> bool:
    return not isinstance(x, (dict, list, tuple))


def tree_flatten_with_path(tree, is_leaf: Callable[[Any], bool] | None = None, _path=()):
    """[(path, leaf)] in JAX order: dict keys sorted, lists/tuples in order; ``None`` is an empty
    subtree unless ``is_leaf`` claims it."""
    if is_leaf is not None and is_leaf(tree):
        return [(_path, tree)]
    if tree is None:
        return []
    if isinstance(tree, dict):
        out = []
        for k in sorted(tree.keys()):
            out += tree_flatten_with_path(tree[k], is_leaf, _path + (k,))
        return out
    if isinstance(tree, (list, tuple)):
        out = []
        for i, v in enumerate(tree):
            out += tree_flatten_with_path(v, is_leaf, _path + (i,))
        return out
    return [(_path, tree)]


def tree_flatten(tree, is_leaf=None):
    return [leaf for _, leaf in tree_flatten_with_path(tree, is_leaf)]


def tree_map(fn, tree, *rest, is_leaf=None):
    """Map over the leaves of ``tree``; the other trees are indexed with the same keys (their
    leaves may be ``None`` or arrays where ``tree`` has scalars, as with flags/transforms)."""
    if is_leaf is not None and is_leaf(tree):
        return fn(tree, *rest)
    if isinstance(tree, dict):
        return {k: tree_map(fn, tree[k], *[r[k] for r in rest], is_leaf=is_leaf) for k in tree}
    if isinstance(tree, (list, tuple)):
        return type(tree)(tree_map(fn, v, *[r[i] for r in rest], is_leaf=is_leaf) for i, v in enumerate(tree))
    return fn(tree, *rest)


def ravel_pytree(tree):
    """(flat 1-D array, unravel) like jax.flatten_util.ravel_pytree."""
    leaves = tree_flatten_with_path(tree)
    shapes = [np.shape(leaf) for _, leaf in leaves]
    sizes = [int(np.prod(s)) if len(s) else 1 for s in shapes]
    flat = np.concatenate([np.ravel(np.asarray(leaf)) for _, leaf in leaves]) if leaves else np.zeros(0)
    paths = [p for p, _ in leaves]

    def unravel(vec):
        vec = np.asarray(vec)
        out: Any = None
        pos = 0
        vals = []
        for shp, n in zip(shapes, sizes):
            chunk = vec[pos:pos + n]
            vals.append(chunk.reshape(shp) if len(shp) else chunk[0])
            pos += n
        return _rebuild(tree, iter(vals))

    return flat, unravel


def _rebuild(template, it):
    if template is None:
        return None
    if isinstance(template, dict):
        # consume in sorted-key order, keep the template's insertion order for iteration
        vals = {k: _rebuild(template[k], it) for k in sorted(template.keys())}
        return {k: vals[k] for k in template}
    if isinstance(template, (list, tuple)):
        return type(template)(_rebuild(v, it) for v in template)
    return next(it)


# --------------------------------------------------------------------------- transforms
# reference :27-54


def bounds_transform(value, bounds, transform_from_canonical=True):
    span = 0.5 * (bounds[1] - bounds[0])
    mean = 0.5 * (bounds[0] + bounds[1])
    if transform_from_canonical:
        return span * value + mean
    t = (value - mean) / span
    if t < -1.:
        t = -1.
    if t > 1.:
        t = 1.
    return t


def log_transform(value, ref_value, transform_from_canonical=True):
    if transform_from_canonical:
        return ref_value[0] * np.exp(value)
    return np.log(value / ref_value[0])


def get_size(x) -> int:
    if isinstance(x, (np.floating, float)):
        return 1
    if isinstance(x, np.ndarray):
        return int(np.size(x))
    raise TypeError(type(x))


def first_deriv_transform(value, transform):        # reference :92-99
    if transform is None:
        return 1.
    if len(transform) == 2:
        return 0.5 * (transform[1] - transform[0])
    if len(transform) == 1:
        return value
    raise ValueError(f"Unexpected transform shape: {transform}")


def second_deriv_transform(value, transform):       # reference :102-109
    if transform is None:
        return 0.
    if len(transform) == 2:
        return 0.
    if len(transform) == 1:
        return value
    raise ValueError(f"Unexpected transform shape: {transform}")


def grad_transform(grad, value, transform):         # reference :87-89
    return first_deriv_transform(value, transform) * grad


def diagonal_hessian_transform(hessian, grad, value, transform):      # reference :112-119
    return hessian * first_deriv_transform(value, transform) ** 2 + grad * second_deriv_transform(value, transform)


def off_diagonal_hessian_transform(hessian, value_ii, value_jj, transform_ii, transform_jj):   # :122-131
    return hessian * first_deriv_transform(value_ii, transform_ii) * first_deriv_transform(value_jj, transform_jj)


def get_opt_bounds(transform):                      # reference :134-138
    if transform is None or len(transform) == 1:
        return [None, None]
    return [-1., 1.]


def transform_from_canonical(value, active_flag, transform):          # reference :141-152
    if active_flag and transform is not None:
        if len(transform) == 2:
            return bounds_transform(value, transform)
        if len(transform) == 1:
            return log_transform(value, transform)
        raise ValueError(f"Unexpected transform shape: {transform}")
    return value


def transform_to_canonical(value, active_flag, transform):            # reference :155-166
    if active_flag and transform is not None:
        if len(transform) == 2:
            return bounds_transform(value, transform, transform_from_canonical=False)
        if len(transform) == 1:
            return log_transform(value, transform, transform_from_canonical=False)
        raise ValueError(f"Unexpected transform shape: {transform}")
    return value


def unpack_elastic_params(params):                  # reference :169-173
    el = params["elastic"]
    return el["E"], el["nu"]


def _transform_is_leaf(x) -> bool:
    """Leaves of a transforms tree are None, or a 1-/2-vector (log ref / bounds)."""
    return x is None or isinstance(x, np.ndarray) or (isinstance(x, (list, tuple)) and len(x) in (1, 2)
                                                      and all(np.isscalar(v) for v in x))


def _expand(values_tree, other_tree, is_leaf):
    """flatten_by_value_size (reference :74-84): one entry of `other_tree` per scalar of `values_tree`."""
    vals = tree_flatten_with_path(values_tree)
    others = dict((p, leaf) for p, leaf in tree_flatten_with_path(other_tree, is_leaf=is_leaf))
    out = []
    for path, v in vals:
        n = get_size(v) if not isinstance(v, (int, np.integer)) else 1
        out += [others[path]] * n
    return out


class Parameters:
    """Handle constitutive model parameters with pytrees (reference :176-401)."""

    def __init__(self, values, active_flags=None, transforms=None) -> None:
        self.values = values
        self._active_flags = active_flags
        self._transforms = transforms

        self._flat_values, self.reconstruct_from_flat = ravel_pytree(values)
        self.num_params = len(self._flat_values)

        flat_with_path = tree_flatten_with_path(values)
        self._names = [str(path[-1]) for path, _ in flat_with_path]
        self._paths = [path for path, _ in flat_with_path]
        self.flat_param_sizes = [get_size(v) if not isinstance(v, (int, np.integer)) else 1 for _, v in flat_with_path]
        self.block_shapes = [(x, y) for x in self.flat_param_sizes for y in self.flat_param_sizes]
        # path of every flat scalar entry (array leaves contribute one path per element)
        self._flat_paths = []
        for (path, _), n in zip(flat_with_path, self.flat_param_sizes):
            self._flat_paths += [path + ((i,) if n > 1 else ()) for i in range(n)]

        if active_flags is not None:
            assert transforms is not None, "transforms must be supplied when active_flags is set"
            self._flat_active_flags = np.array(
                _expand(values, active_flags, is_leaf=lambda x: isinstance(x, (bool, np.bool_))), dtype=bool)
            self.num_active_params = int(np.sum(self._flat_active_flags))
            self.active_idx = np.arange(self.num_params)[self._flat_active_flags]
            self.model_active_params_jacobian = partial(self._active_params_jacobian, active_idx=self.active_idx)
            self.qoi_active_params_jacobian = partial(self._active_params_jacobian, num_eqns=1,
                                                      active_idx=self.active_idx)
            self._flat_transforms = _expand(values, transforms, is_leaf=_transform_is_leaf)
            self._expanded_flat_transforms = self._flat_transforms
            self._flat_active_transforms = [self._flat_transforms[ii] for ii in self.active_idx]
            self.opt_bounds = np.array([get_opt_bounds(t) for t in self._flat_active_transforms])
            self.get_params_pytree_from_flat_canonical_active = partial(
                self._get_params_pytree_from_flat_canonical_active,
                flat_values=self._flat_values, reconstruct_from_flat=self.reconstruct_from_flat,
                active_idx=self.active_idx, active_flags=active_flags, transforms=transforms)
        else:
            assert active_flags == transforms
            self.num_active_params = 0

    # ------------------------------------------------------------------ reference :262-266
    def set_rotation_matrix(self, rotation_matrix) -> None:
        self.values["rotation matrix"] = rotation_matrix
        self._flat_values, _ = ravel_pytree(self.values)

    # ------------------------------------------------------------------ reference :269-280
    def set_active_values(self, values, are_canonical: bool = True) -> None:
        if are_canonical:
            self.values = self._map3(transform_from_canonical, values)
        else:
            self.values = values

    def _map3(self, fn, values):
        """tree_map(fn, values, active_flags, transforms) with element-wise handling of array leaves."""
        flat, unravel = ravel_pytree(values)
        out = np.array([fn(v, a, t) for v, a, t in zip(flat, self._flat_active_flags, self._flat_transforms)],
                       dtype=flat.dtype)
        return unravel(out)

    # ------------------------------------------------------------------ reference :283-303
    def set_active_values_from_flat(self, flat_active_values, are_canonical: bool = True,
                                    is_complex: bool = False) -> None:
        if is_complex:
            updated = np.array(self._flat_values, dtype=complex)
        else:
            updated = np.array(self._flat_values)
        updated[self.active_idx] = flat_active_values
        self.set_active_values(self.reconstruct_from_flat(updated), are_canonical)

    # ------------------------------------------------------------------ reference :306-318
    def flat_active_values(self, return_canonical: bool = False):
        flat_values, _ = ravel_pytree(self.values)
        if return_canonical:
            return np.array([transform_to_canonical(v, a, t) for v, a, t in
                             zip(flat_values, self._flat_active_flags, self._flat_transforms)])[self.active_idx]
        return np.asarray(flat_values[self.active_idx])

    def get_active_from_flat(self, pytree):             # reference :321-323
        flat, _ = ravel_pytree(pytree)
        return flat[self.active_idx]

    # ------------------------------------------------------------------ reference :326-331 (in place)
    def transform_grad(self, grad) -> None:
        active = self.get_active_from_flat(self.values)
        for ii in range(self.num_active_params):
            grad[ii] = grad_transform(grad[ii], active[ii], self._flat_active_transforms[ii])

    # ------------------------------------------------------------------ reference :334-357 (in place)
    def transform_hessian(self, hessian, grad) -> None:
        active = self.get_active_from_flat(self.values)
        n = self.num_active_params
        for ii in range(n):
            for jj in range(n):
                if ii == jj:
                    hessian[ii, ii] = diagonal_hessian_transform(hessian[ii, ii], grad[ii], active[ii],
                                                                 self._flat_active_transforms[ii])
                elif ii < jj:
                    hessian[ii, jj] = off_diagonal_hessian_transform(
                        hessian[ii, jj], active[ii], active[jj],
                        self._flat_active_transforms[ii], self._flat_active_transforms[jj])
                else:
                    hessian[ii, jj] = hessian[jj, ii]

    def compute_mixed_block_shapes(self, num_eqs) -> None:      # reference :360-365
        self.mixed_block_shapes = [(x, y) for x in num_eqs for y in self.flat_param_sizes]

    # ------------------------------------------------------------------ reference :368-381
    @staticmethod
    def _active_params_jacobian(jacobian, num_eqns, active_idx):
        """``jacobian`` is a pytree shaped like ``values`` whose leaves are (num_eqns, *leaf_shape)."""
        leaves = tree_flatten(jacobian)
        arr = np.hstack([np.asarray(x).reshape(num_eqns, -1) for x in leaves])
        return arr[:, active_idx]

    def scalar_active_params_jacobian(self, jacobian):
        return self._active_params_jacobian(jacobian, 1, self.active_idx)

    # ------------------------------------------------------------------ reference :384-401
    @staticmethod
    def _get_params_pytree_from_flat_canonical_active(flat_canonical_active, flat_values, reconstruct_from_flat,
                                                      active_idx, active_flags, transforms):
        flat_values = np.array(flat_values)
        for idx, v in zip(active_idx, flat_canonical_active):
            flat_values[idx] = v
        pytree = reconstruct_from_flat(flat_values)
        flags = _expand(pytree, active_flags, is_leaf=lambda x: isinstance(x, (bool, np.bool_)))
        trans = _expand(pytree, transforms, is_leaf=_transform_is_leaf)
        flat, unravel = ravel_pytree(pytree)
        return unravel(np.array([transform_from_canonical(v, a, t) for v, a, t in zip(flat, flags, trans)]))

    # ------------------------------------------------------------------ additions for the HIP facade
    def flat_paths(self):
        """Key path of every flat scalar entry, in flat order (used to map kernel sensitivities)."""
        return list(self._flat_paths)

    def active_paths(self):
        return [self._flat_paths[i] for i in getattr(self, "active_idx", [])]
