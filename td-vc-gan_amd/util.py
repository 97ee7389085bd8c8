"""Device-side counterparts of the reference's util/__init__.py helpers that sit directly before the train step."""
import math

import torch

from . import _lib as L


def f0_to_excitation(f0, step_size, sampling_rate=16000, linear=True, noise=None, start_phase=None):
    """Sine + noise excitation from a frame-level F0 track, on the GPU (reference: util/__init__.py:22-50, same name and
    arguments). f0: [B, 1, n_frames] in Hz, 0 = unvoiced; returns [B, 1, (n_frames - 1) * step_size].

    The reference draws its random numbers inside the function; here they may be passed in (`noise` = (n_voiced, n_unvoiced)
    standard-normal tensors [B, 1, T], `start_phase` = 1-element tensor in radians) so that both sides of a parity test use
    the same draw; by default they are drawn on the device with torch's generator.
    """
    if not f0.is_cuda:
        raise L.TdvcError('f0_to_excitation: the HIP path needs a CUDA/ROCm tensor (there is no CPU fallback)')
    f0 = f0.contiguous().float()
    B, one, nf = f0.shape
    if one != 1 or nf < 2:
        raise ValueError('f0 must be [B, 1, n_frames >= 2]')
    T = (nf - 1) * int(step_size)
    dev = f0.device
    if noise is None:
        noise = (torch.randn(B, 1, T, device=dev), torch.randn(B, 1, T, device=dev))
    nv, nu = (t.contiguous().float() for t in noise)
    if start_phase is None:
        start_phase = torch.rand(1, device=dev) * (2 * math.pi)
    start_phase = start_phase.reshape(1).contiguous().float()
    exc = torch.empty(B, 1, T, dtype=torch.float32, device=dev)
    L.check(L.lib().tdvc_f0_to_excitation(f0.data_ptr(), nv.data_ptr(), nu.data_ptr(), start_phase.data_ptr(), exc.data_ptr(),
                                          B, nf, int(step_size), float(sampling_rate), int(bool(linear)),
                                          torch.cuda.current_stream(dev).cuda_stream))
    return exc
