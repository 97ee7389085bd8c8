"""tdvc-mi355x: MI355X-native (gfx950) implementation of TD-VC-GAN's G+D train-step path.

Drop-in surface (same names / signatures / state_dict keys as the reference):
    Generator, CollaborativeMultibandDiscriminator, ConditionalInstanceNorm, LatentClassifier,
    losses.{multiscale_spec_loss, multiscale_feat_loss, contrastive_loss}, TrainStep.

The HIP library (csrc/ -> libtdvc_hip.so, C ABI in include/tdvc.h) is loaded lazily on the first
operator call and the load fails loudly when it is missing: there is no CPU or ATen fallback in
the product path. The directory name contains hyphens; import it as `import tdvc_amd` (alias
module at the repo root) or `importlib.import_module('td-vc-gan_amd')`.
"""
__version__ = '0.1.0'

from . import synth  # noqa: F401  (numpy/torch only, no GPU)


def __getattr__(name):
    # heavy submodules on demand, so that `synth` stays importable without torch.cuda / the .so
    import importlib
    if name in ('modules', 'losses', 'ops', 'arena', 'train_step', 'parallel', 'hparams', 'util', 'infer', 'ssl_encoder', '_lib'):
        return importlib.import_module(f'{__name__}.{name}')
    if name in ('Generator', 'CollaborativeMultibandDiscriminator', 'ConditionalInstanceNorm', 'LatentClassifier', 'Discriminator'):
        return getattr(importlib.import_module(f'{__name__}.modules'), name)
    if name in ('TrainStep', 'StepConfig'):
        return getattr(importlib.import_module(f'{__name__}.train_step'), name)
    raise AttributeError(name)
