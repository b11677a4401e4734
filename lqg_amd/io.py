"""Tracking-data loader — mirrors lqg/io.py:45-98 (`load_tracking_data`), the format either side of the hot path:
the Bonnen et al. (2015) `data.mat` (fields `sigma[n]`, `target[n, S]`, `response[n, S]`) becomes the
`(n_conditions, n_trials, T, 2)` float32 array of (target, response) trajectories that `log_likelihood` /
`lqg_amd.infer` consume, plus the sorted blob widths in arcmin.  Host-side numpy; feed the GPU with
`torch.as_tensor(data[c], device="cuda")`.
"""
import os

import numpy as np
import scipy.io as spio

ARCMIN_PER_PIXEL = 1.32                                          # lqg/io.py:58


def loadmat(filename):
    """lqg/io.py:9-17: MATLAB file -> dict; nested structs become nested dicts."""
    def plain(v):
        if hasattr(v, "_fieldnames"):
            return {k: plain(getattr(v, k)) for k in v._fieldnames}
        return v

    raw = spio.loadmat(filename, struct_as_record=False, squeeze_me=True)
    return {k: plain(v) for k, v in raw.items()}


def load_tracking_data(delay=12, clip=120, subtract_mean=True, data_path="data/"):
    """Load tracking data from Bonnen et al. (2015)  (lqg/io.py:45-98).

    Args:
        delay: temporal delay (steps) between target and response: response[t + delay] is paired with target[t]
        clip: drop the first `clip` time steps
        subtract_mean: subtract each trial's temporal mean from both trajectories
        data_path: directory holding data.mat

    Returns:
        (data float32 [n_conditions, n_trials, T, 2], sigmas [n_conditions])
    """
    mat = loadmat(os.path.join(data_path, "data.mat"))
    width = (mat["sigma"] * ARCMIN_PER_PIXEL).round()           # blob width per trial, arcmin
    sigmas = np.unique(width)
    target = mat["target"].astype(np.float32)
    response = mat["response"].astype(np.float32)
    if delay:
        target, response = target[:, clip:-delay], response[:, clip + delay:]
    else:
        target, response = target[:, clip:], response[:, clip:]
    if subtract_mean:
        target = target - np.mean(target, axis=1, keepdims=True)
        response = response - np.mean(response, axis=1, keepdims=True)
    pair = np.stack([target, response], axis=-1)                # [trial, T, 2]
    groups = [pair[width == w] for w in sigmas]
    if len({g.shape[0] for g in groups}) != 1:
        raise ValueError("load_tracking_data: conditions have different trial counts: "
                         f"{[g.shape[0] for g in groups]}")
    data = np.stack(groups)                                     # [condition, trial, T, 2]
    data = data - data[:, :, :1, :1]                            # every trial starts with the target at 0
    return data, sigmas
