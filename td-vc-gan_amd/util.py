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


def load_possible(model, state_dict):
    """Permissive checkpoint load (reference: util/__init__.py:64-89, same name / arguments / return value): tensors whose
    key and shape match are taken over, tensors whose key matches but whose shape differs are copied on their common
    leading slices, everything else is reported. Returns the reference's message dict
    {'matched', 'mismatched_size', 'unmatched_keys', 'missing_keys'}.

    One deliberate difference: the reference only REBINDS matched entries in a throw-away dict and relies on the failed
    strict `load_state_dict` that precedes it in train.py:58-68 (which copies every shape-compatible tensor before it
    raises) to have loaded them; here matched tensors are copied in place as well, so the function is correct on its own.
    Arena-backed models refresh their effective weights afterwards."""
    own = model.state_dict()
    messages = {'matched': [], 'mismatched_size': [], 'unmatched_keys': [], 'missing_keys': []}
    with torch.no_grad():
        for k, v in state_dict.items():
            if k not in own:
                messages['unmatched_keys'].append(k)
                continue
            dst = own[k]
            if v.shape == dst.shape:
                dst.copy_(v)
                messages['matched'].append(k)
            elif v.dim() == dst.dim():
                sl = tuple(slice(0, min(a, b)) for a, b in zip(dst.shape, v.shape))
                dst[sl] = v[sl].to(dst.dtype)
                messages['mismatched_size'].append(k)
            else:
                raise RuntimeError(f'load_possible: {k} has {v.dim()} dims in the checkpoint and {dst.dim()} in the model')
    for k in own:
        if k not in state_dict:
            messages['missing_keys'].append(k)
    arena = getattr(model, '_arena', None)
    if arena is not None:
        arena.materialize()
    return messages


def load_model(model, path):
    """train.py:58-68 — strict load, falling back to `load_possible`. The file is read with weights_only=True: nothing
    from a checkpoint is ever executed."""
    state_dict = torch.load(path, map_location='cpu', weights_only=True)
    try:
        model.load_state_dict(state_dict)
        return None
    except RuntimeError:
        print(f'Warning: default loading for {path} failed. Trying permisive load')
        messages = load_possible(model, state_dict)
        for kind, keys in messages.items():
            if kind != 'matched':
                for k in keys:
                    print(f'{kind}: {k}')
        return messages
