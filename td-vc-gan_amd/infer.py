"""Forward-only conversion path: what generate_with_target.py:54-184 does around the generator call, with the
same kernels as the train step — checkpoint interchange (`step{E}-G.pt` / `latest-G.pt` are plain state_dicts with the
reference's keys), log-F0 mean shift towards the target speaker, excitation on the device, one whole-utterance forward.

F0 extraction (CREPE, util/crepe.py) and audio I/O stay with the caller: F0 tracks are inputs here.
"""
import torch

from . import modules as M
from .util import f0_to_excitation

SEGMENT_MULT = 320       # product of the decoder ratios (data/dataset.py:41): utterances are padded to a multiple of it
F0_HOP = 64


def build_generator(gen_cfg, num_spk, device):
    """Generator from the `model.generator` section of a reference config (train.py:125-138 argument order)."""
    g = gen_cfg
    nl, wn, cond = g['norm_layer'], g['weight_norm'], g['conditioning']
    G = M.Generator(list(g['decoder_ratios']), list(g['decoder_channels']), g['num_bottleneck_layers'], num_spk,
                    g['conditional_dim'], g['content_dim'], g['num_res_blocks'], g['num_enc_layers'], g['encoder_model'],
                    norm_layer=(nl['bottleneck'], nl['encoder'], nl['decoder']),
                    weight_norm=(wn['bottleneck'], wn['encoder'], wn['decoder']),
                    bot_cond=cond['bottleneck'], enc_cond=cond['encoder'], dec_cond=cond['decoder'])
    G.ensure_arena(device)
    return G


def load_generator_checkpoint(G, path):
    """Load a reference checkpoint (a state_dict saved by train.py:597-608). Nothing from the file is executed."""
    sd = torch.load(path, map_location='cpu', weights_only=True)
    G.load_state_dict(sd)      # copies into the arena-backed parameters in place
    return G


def shift_f0(f0_src, f0_tgt):
    """log-F0 mean shift of the source contour towards the target speaker (generate_with_target.py:135-160,
    train.py:245-251). f0: [B, 1, n] in Hz, 0 = unvoiced. Element-wise pre-processing on a few hundred frames."""
    def mu(f):
        v = f > 0
        return (v * torch.log(f + 1e-6)).sum(-1, keepdim=True) / (v.sum(-1, keepdim=True) + 1e-6)
    out = torch.zeros_like(f0_src)
    voiced = f0_src > 0
    out[voiced] = torch.exp(torch.log(f0_src + 1e-6) + mu(f0_tgt) - mu(f0_src))[voiced]
    return out


@torch.no_grad()
def convert(G, signal, c_tgt, f0_conv, noise=None, start_phase=None):
    """signal [1, 1, T] (T a multiple of 320, >= 8320), c_tgt one-hot [1, num_spk], f0_conv [1, 1, T/64 + 1] -> converted
    signal [1, 1, T]. One forward of the whole utterance, as the reference does (test.max_segment = 71680 samples)."""
    T = signal.shape[-1]
    if T % SEGMENT_MULT:
        raise ValueError(f'utterance length {T} is not a multiple of {SEGMENT_MULT} (pad it as data/dataset.py:158-163 does)')
    exc = f0_to_excitation(f0_conv, F0_HOP, noise=noise, start_phase=start_phase)
    if exc.shape[-1] != T:
        raise ValueError(f'F0 track gives {exc.shape[-1]} excitation samples for a {T}-sample utterance')
    return G(signal, c_tgt, c_var=exc)
