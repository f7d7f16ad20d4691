"""Mean / STD frames of a video - mirror of modules/video_processing.py:161-262 on the HIP backend.

The reference streams frames out of `cv.VideoCapture` one at a time and updates three whole-image NumPy
arrays per frame (video_processing.py:199-208). Here the decoded frames are uploaded in batches and folded
into the device-resident float64 state by `hm_welford_update`, which reads and writes the state once per
batch (not once per frame); the frame order and the per-frame arithmetic are the reference's.

Video decoding stays with the caller: OpenCV is not a dependency of this package, so `welford_algorithm`
takes any iterable of uint8 (H, W, C) frames - e.g. `frames_from_capture(cv.VideoCapture(path))` - or a list
of such iterables (the reference's "all videos of a directory" mode, :193-195).

Deviation K (SURVEY.md 3.4 style): `if ICRF:` at :200 raises for an ndarray; the intended `ICRF is not None` is used.
"""
from __future__ import annotations

from typing import Iterable, Iterator, Optional, Sequence, Union

import numpy as np
import torch

from . import engine


def _engine_for(device: torch.device):
    """The HIP library for a GPU device; the host build of the same ABI for device="cpu" - an explicit choice (the reference's NumPy
    slot), never a fallback: the default device is the current GPU and the calls below raise without one."""
    if device.type == "cuda":
        return engine
    from .measurand import _HOST_ENGINE
    return _HOST_ENGINE


FRAMES_PER_LAUNCH = 32      # = HM_MAX_FRAMES: state traffic per element-frame = 32 / 32 = 1 byte with std (measured best)


def frames_from_capture(capture) -> Iterator[np.ndarray]:
    """general_functions.video_frame_generator (modules/general_functions.py:226-251) for an already opened
    cv.VideoCapture-like object (`isOpened()`, `read() -> (ok, frame)`, `release()`)."""
    if not capture.isOpened():
        raise ValueError("Unable to open video file")
    try:
        while True:
            ok, frame = capture.read()
            if not ok:
                break
            yield frame
    finally:
        capture.release()


def _as_device_frame(frame, device) -> torch.Tensor:
    if isinstance(frame, torch.Tensor):
        t = frame
    else:
        t = torch.from_numpy(np.ascontiguousarray(frame))
    if t.dtype != torch.uint8:
        raise TypeError(f"video frames must be uint8, got {t.dtype}")
    if t.dim() == 2:
        t = t[..., None]
    if torch.device(device).type == "cpu":
        return t.clone()                      # (a capture may hand out the same buffer again; a batch of frames is pending at a time)
    return t.to(device, non_blocking=True)


def welford_algorithm(frame_sources: Union[Iterable, Sequence[Iterable]], ICRF: Optional[np.ndarray] = None,
                      use_std: Optional[bool] = False, device=None, frames_per_launch: int = FRAMES_PER_LAUNCH,
                      as_numpy: bool = True):
    """Mean frame and standard-deviation-of-the-mean frame over all frames (video_processing.py:161-219).

    Args:
        frame_sources: an iterable of uint8 (H, W, C) frames, or a list of such iterables (several videos folded
            into one result). A frame of None ends a source (the reference's generator protocol, :192-193).
        ICRF: (BITS, NUM_OF_CHS) float64 inverse camera response; frames are linearized on load when given.
        use_std: whether to compute the standard deviation frame.
        device: the GPU to compute on (default: the current one), or "cpu" for the host build (libhdrmerge_host.so).
    Returns:
        {'mean': uint8 (H, W, C), 'std': uint8 (H, W, C) or None} - NumPy arrays (or device tensors with as_numpy=False).
    """
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    eng = _engine_for(device)
    if frames_per_launch < 1:
        raise ValueError("frames_per_launch must be positive")
    if isinstance(frame_sources, (list, tuple)) and frame_sources and not _looks_like_frame(frame_sources[0]):
        sources = list(frame_sources)
    else:
        sources = [frame_sources]

    mean = m2 = None
    count = 0
    pending = []

    def flush():
        nonlocal count
        if pending:
            count = eng.welford_update(pending, count, mean, m2, ICRF)
            pending.clear()

    for source in sources:
        for frame in source:
            if frame is None:
                break
            t = _as_device_frame(frame, device)
            if mean is None:
                mean = torch.zeros(t.shape, dtype=torch.float64, device=device)              # :181
                m2 = torch.zeros(t.shape, dtype=torch.float64, device=device) if use_std else None   # :184
            elif t.shape != mean.shape:
                raise ValueError(f"frame shape {tuple(t.shape)} differs from the first frame's {tuple(mean.shape)}")
            pending.append(t)
            if len(pending) == frames_per_launch:
                flush()
    flush()
    if mean is None:
        raise ValueError("no frames to process")
    if use_std and count < 2:
        raise ValueError("the standard deviation frame needs at least two frames")
    out_mean, out_std = eng.welford_finalize(mean, m2, count)
    if as_numpy:
        return {"mean": out_mean.cpu().numpy(), "std": None if out_std is None else out_std.cpu().numpy()}
    return {"mean": out_mean, "std": out_std}


def _looks_like_frame(x) -> bool:
    return x is None or (isinstance(x, (np.ndarray, torch.Tensor)) and x.ndim in (2, 3))


def save_result(ret: dict, video_path) -> None:
    """The file naming of process_video (video_processing.py:222-236): '<video>.mean.tif' / '<video>.std.tif'
    next to the video, written with cv.imwrite's conventions."""
    from pathlib import Path
    from . import tiff_io
    video_path = Path(video_path)
    for key in ret:
        if ret[key] is not None:
            img = ret[key].cpu().numpy() if isinstance(ret[key], torch.Tensor) else ret[key]
            tiff_io.imwrite(video_path.parent.joinpath(video_path.name.replace(".avi", f".{key}.tif")), img)


def process_video(frame_source, video_path, ICRF: Optional[np.ndarray] = None, use_std: Optional[bool] = True):
    """video_processing.py:222-236 for an already opened frame source."""
    ret = welford_algorithm(frame_source, ICRF, use_std)
    save_result(ret, video_path)
    return ret
