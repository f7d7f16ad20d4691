"""ICRF calibration by exposure linearity - mirror of modules/ICRF_calibration_exposure.py on the HIP backend.

The reference minimises, per channel, an energy function with SciPy's differential-evolution solver; every
evaluation maps the (X, Y, N) channel stack through a candidate ICRF and reduces an (X, Y, N, N) array of
pairwise relative differences on the host (ICRF_calibration_exposure.py:66-201), one candidate at a time.
Here the stack stays on the device and `hm_linearity_energy` evaluates a whole POPULATION of candidates in one
launch (grid = pixel chunks x frame pairs x candidates); the solver itself stays SciPy's, on the host, driven
through its `vectorized=True` interface (one launch per generation) or, for the reference's exact update
order, one candidate per call.

The candidate ICRF (mean + PCA product, shift, range / monotonicity rejection: :22-45,:166-179) is 256 numbers
per candidate and is formed on the host with NumPy, like the reference.
"""
from __future__ import annotations

import inspect
from typing import Optional, Sequence

import numpy as np
import torch
from scipy.optimize._differentialevolution import DifferentialEvolutionSolver   # same access as the reference (:7)

from . import engine
from . import settings as gs


def _inverse_camera_response_function(mean_ICRF, PCA_array, PCA_params, use_mean_ICRF):
    """:22-45. PCA_params may be one vector (n_params,) or a batch (n_candidates, n_params); returns (256,) or
    (n_candidates, 256)."""
    p = np.asarray(PCA_params, dtype=np.float64)
    single = p.ndim == 1
    p = np.atleast_2d(p)
    if not use_mean_ICRF:
        base = np.linspace(0, 1, gs.BITS)[None, :] ** p[:, :1]
        out = np.stack([base[b] + np.matmul(PCA_array, p[b, 1:]) for b in range(p.shape[0])])
    else:
        out = np.stack([mean_ICRF + np.matmul(PCA_array, p[b]) for b in range(p.shape[0])])
    return out[0] if single else out


def candidate_icrfs(PCA_params, mean_ICRF, PCA_array, use_mean_ICRF=True):
    """Candidate ICRFs as the energy function sees them (:165-179): shifted so that ICRF[-1] = 1 and ICRF[0] = 0, and the
    per-candidate verdict of the range and strict-monotonicity tests. -> (icrfs (B, 256), valid (B,) bool)."""
    icrfs = np.atleast_2d(_inverse_camera_response_function(mean_ICRF, PCA_array, PCA_params, use_mean_ICRF)).copy()
    icrfs += (1 - icrfs[:, -1])[:, None]                                         # :166
    icrfs[:, 0] = 0                                                              # :167
    valid = ~((icrfs.max(axis=1) > 1) | (icrfs.min(axis=1) < 0))                 # :173-175
    valid &= np.all(icrfs[:, 1:] > icrfs[:, :-1], axis=1)                        # :177-179
    return icrfs, valid


def analyze_linearity(image_value_stack: torch.Tensor, image_std_stack: Optional[torch.Tensor], ICRF_ch, lower: int, upper: int,
                      use_relative: bool, exposure_values):
    """:66-145 for a uint8 DN stack seen through one ICRF: the N(N-1)/2 pair results (device tensor) in
    np.triu_indices(N, 1) order. `lower` / `upper` are DN limits (the energy function maps them through the ICRF, :181-182)."""
    _, pairs = _engine_for(image_value_stack).linearity_energy(image_value_stack, image_std_stack, _host(exposure_values),
                                                               np.asarray(ICRF_ch)[None], lower, upper, None, use_relative, return_pairs=True)
    return pairs[0]


def _host(x):
    return x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x, dtype=np.float64)


def _engine_for(stack: torch.Tensor):
    """The HIP library for a device-resident stack; the host build of the same ABI for a host stack (initialize_channel_image_stacks(...,
    device="cpu")). The reference makes this choice once per calibration - `calibration(..., use_cupy)` switches the module's array
    library (modules/ICRF_calibration_exposure.py:17, :317-319) and the stacks are created in it; here the choice is the `device` the
    caller gives initialize_channel_image_stacks (default: the current GPU - it raises without one) and every later call follows the
    stacks' placement. Never a fallback: a device stack never computes on the host."""
    if not isinstance(stack, torch.Tensor) or stack.is_cuda:        # (anything but a tensor: engine raises the TypeError)
        return engine
    from .measurand import _HOST_ENGINE
    return _HOST_ENGINE


def energy_function_batch(PCA_params, mean_ICRF, PCA_array, image_value_stack, image_std_stack, lower, upper, use_mean,
                          exposure_values) -> np.ndarray:
    """Energies of a batch of candidates. PCA_params: (n_candidates, n_params). One device launch."""
    icrfs, valid = candidate_icrfs(PCA_params, mean_ICRF, PCA_array, use_mean)
    if not valid.any():
        return np.full(len(valid), np.inf)
    e = _engine_for(image_value_stack).linearity_energy(image_value_stack, image_std_stack, _host(exposure_values), icrfs, int(lower), int(upper),
                                valid, True)
    return e.cpu().numpy()


def _energy_function(PCA_params, mean_ICRF, PCA_array, image_value_stack, image_std_stack, lower, upper, use_mean,
                     exposure_values):
    """:148-201, the reference's signature: one candidate -> float. With a (n_params, S) array (SciPy's
    `vectorized=True` calling convention) -> (S,) energies from one launch."""
    p = np.asarray(PCA_params, dtype=np.float64)
    if p.ndim == 2:
        return energy_function_batch(p.T, mean_ICRF, PCA_array, image_value_stack, image_std_stack, lower, upper, use_mean,
                                     exposure_values)
    return float(energy_function_batch(p[None], mean_ICRF, PCA_array, image_value_stack, image_std_stack, lower, upper, use_mean,
                                       exposure_values)[0])


def interpolate_ICRF(ICRF_array, datapoints: Optional[int] = None):
    """:204-216: resample (DATAPOINTS, C) to (BITS, C) by linear interpolation when the sizes differ."""
    ICRF_array = np.asarray(ICRF_array, dtype=np.float64)
    datapoints = ICRF_array.shape[0] if datapoints is None else datapoints
    if gs.BITS == datapoints:
        return ICRF_array
    x_new = np.linspace(0, 1, num=gs.BITS)
    x_old = np.linspace(0, 1, num=datapoints)
    out = np.zeros((gs.BITS, ICRF_array.shape[1]), dtype=float)
    for c in range(ICRF_array.shape[1]):
        out[:, c] = np.interp(x_new, x_old, ICRF_array[:, c])
    return out


def initialize_channel_image_stacks(frames: Sequence, exposures: Sequence[float], stds: Optional[Sequence] = None,
                                    data_spacing=150, device=None):
    """:219-284 for frames already in memory (uint8 (H, W, C) arrays / tensors): sort by exposure, thin the pixels with
    `data_spacing` (int or (x_step, y_step); plain strided selection) and stack every channel to (X, Y, N).
    -> (channel value stacks [C x uint8 (X, Y, N) device tensors], channel std stacks or [None]*C, exposures ndarray)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    x_step, y_step = data_spacing if isinstance(data_spacing, tuple) else (data_spacing, data_spacing)
    order = np.argsort(np.asarray(exposures, dtype=np.float64), kind="stable")
    t = np.asarray(exposures, dtype=np.float64)[order]

    def thin(img, dtype):
        a = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))
        return a[::x_step, ::y_step].to(device=device, dtype=dtype)

    vals = torch.stack([thin(frames[i], torch.uint8) for i in order], dim=-1)            # (X, Y, C, N)
    value_stacks = [vals[:, :, c, :].contiguous() for c in range(vals.shape[2])]
    if stds is not None:
        sds = torch.stack([thin(stds[i], torch.float64) for i in order], dim=-1)
        std_stacks = [sds[:, :, c, :].contiguous() for c in range(sds.shape[2])]
    else:
        std_stacks = [None] * vals.shape[2]
    return value_stacks, std_stacks, t


def solve_channel(mean_ICRF_array, PCA_array, image_value_stack, image_std_stack, exposure_values,
                  lower_PCA_limit: float, upper_PCA_limit: float, use_mean_ICRF: bool = True,
                  data_limits=(gs.LOWER_LIN_LIM, gs.UPPER_LIN_LIM), energy_limit: float = 0.0, seed=7,
                  max_iterations: int = 1000, vectorized: bool = True, popsize: int = 15, channel: int = 0, verbose: bool = False):
    """The per-channel solve of calibration() (:329-369): SciPy's DifferentialEvolutionSolver with the reference's
    settings. vectorized=True evaluates each generation's population in ONE launch (SciPy then uses deferred updating);
    vectorized=False keeps the reference's immediate updating and evaluates one candidate per launch.
    -> (ICRF of the channel (256,), final energy, iterations)."""
    PCA_array = np.asarray(PCA_array, dtype=np.float64)
    n_params = PCA_array.shape[1] + (0 if use_mean_ICRF else 1)
    limits, x0 = [], []
    if not use_mean_ICRF:
        limits.append([1, 8])                                                    # :309-311
        x0.append(3)
    for _ in range(PCA_array.shape[1]):
        limits.append([lower_PCA_limit, upper_PCA_limit])                        # :313-315
        x0.append(0)
    assert len(limits) == n_params
    args = (mean_ICRF_array, PCA_array, image_value_stack, image_std_stack, data_limits[0], data_limits[1], use_mean_ICRF,
            exposure_values)
    extra = dict(vectorized=True, updating="deferred") if vectorized else {}
    # the reference passes seed= (SciPy 1.14, its Pipfile.lock); SciPy >= 1.15 renamed the argument to rng=
    extra["rng" if "rng" in inspect.signature(DifferentialEvolutionSolver.__init__).parameters else "seed"] = seed
    number_of_iterations = 0
    func_value = np.inf
    with DifferentialEvolutionSolver(_energy_function, limits, args=args, strategy="currenttobest1bin", tol=0.01, x0=x0,
                                     mutation=(0, 1.95), recombination=0.4, init="sobol", popsize=popsize,
                                     **extra) as solver:                          # :344-347
        for step in solver:
            number_of_iterations += 1
            try:
                step = next(solver)          # as written (:351): every pass of the loop advances two generations
            except StopIteration:
                pass
            func_value = step[1]
            if verbose and number_of_iterations % 20 == 0:
                print(f"Channel {channel} value: {func_value} on step {number_of_iterations}")
            if solver.converged() or number_of_iterations == max_iterations or func_value < energy_limit:   # :356
                break
        result = solver.x
    icrf = _inverse_camera_response_function(mean_ICRF_array, PCA_array, result, use_mean_ICRF)
    return icrf, float(func_value), number_of_iterations


def calibration(mean_ICRFs: Sequence, PCA_arrays: Sequence, channel_image_value_stacks, channel_image_std_stacks, exposure_values,
                lower_PCA_limit: float, upper_PCA_limit: float, initial_function=None,
                data_limits=(gs.LOWER_LIN_LIM, gs.UPPER_LIN_LIM), energy_limit: float = 0.0, rng_seed: int = 7,
                vectorized: bool = True, max_iterations: int = 1000, popsize: int = 15):
    """calibration() (:287-405) from arrays: per-channel mean ICRF (or `initial_function`) and PCA basis, the channel stacks
    of initialize_channel_image_stacks. The channels are solved one after another on this process's GPU (the reference
    forks one joblib worker per channel, :383; with one process per GPU, give each rank a channel instead).
    -> (ICRF (BITS, C) interpolated, final energies (C,))."""
    C = len(channel_image_value_stacks)
    use_mean_ICRF = initial_function is None
    results, energies = [], np.zeros(C)
    for c in range(C):
        base = mean_ICRFs[c] if use_mean_ICRF else initial_function
        icrf_c, energies[c], _ = solve_channel(base, PCA_arrays[c], channel_image_value_stacks[c], channel_image_std_stacks[c],
                                               exposure_values, lower_PCA_limit, upper_PCA_limit, use_mean_ICRF, data_limits,
                                               energy_limit, rng_seed + c, max_iterations, vectorized, popsize, c)
        results.append(icrf_c)
    ICRF = np.stack(results, axis=1)
    ICRF += (1 - ICRF[-1, :])[None, :]                                           # :390
    ICRF[0, :] = 0                                                               # :391
    ICRF[ICRF < 0] = 0                                                           # :395-396
    ICRF[ICRF > 1] = 1
    return interpolate_ICRF(ICRF), energies
