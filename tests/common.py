"""Shared test helpers: model construction from the golden shape tables + deterministic weights."""
import importlib
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

G_ARGS = dict(decoder_ratios=[10, 8, 2, 2], decoder_channels=[256, 128, 64, 32, 16], num_bottleneck_layers=0,
              num_classes=16, conditional_dim=128, content_dim=128, num_res_blocks=3, num_enc_layers=16,
              encoder_model='conv', norm_layer=(None, None, None), weight_norm=('weight_norm',) * 3,
              bot_cond='target', enc_cond=None, dec_cond='target', output_content_emb=True)
D_ARGS = dict(num_disc=3, num_classes=16, num_layers=4, num_channels_base=16, num_channel_mult=4,
              downsampling_factor=4, conditional_dim=128, conditional='target')


def pkg():
    return importlib.import_module('td-vc-gan_amd')


def shapes(which):
    return json.load(open(os.path.join(GOLDEN, f'shapes_{which}.json')))


def filled_sd(which):
    return pkg().synth.fill_state_dict(shapes(which))


def build_models(dev):
    M = pkg().modules
    G = M.Generator(**{**G_ARGS, 'decoder_channels': list(G_ARGS['decoder_channels'])})
    D = M.CollaborativeMultibandDiscriminator(**D_ARGS)
    G.load_state_dict(filled_sd('G')); D.load_state_dict(filled_sd('D'))
    G.ensure_arena(dev); D.ensure_arena(dev)
    return G, D


def to_dev(batch, dev):
    return {k: v.to(dev) for k, v in batch.items()}


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
