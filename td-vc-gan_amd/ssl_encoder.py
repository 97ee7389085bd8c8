"""SSL (WavLM-conditioned) content encoder on the HIP path — SURVEY §8f-4. Mirrors the reference's
model/ssl_encoder.py: `WN` (:17-90, gated WaveNet stack), `Encoder` (:93-116, pre 1x1 -> WN -> proj 1x1) and
`SSLEncoder` (:118-148), with the reference's attribute names so that the state_dict keys interchange
(`encoder.pre.weight`, `encoder.enc.in_layers.{i}.weight_g/_v/bias`, `encoder.enc.res_skip_layers.{i}...`,
`encoder.proj.*`).

The frozen WavLM-Large feature extractor itself is OUT of scope of the HIP path (third-party model, stock PyTorch, and
its checkpoint `wavlm/WavLM-Large.pt` does not ship with the reference): `SSLEncoder` takes it as an injected module
`cmodel` (anything with the reference's `extract_features(wave)[0] -> [B, T', 1024]`), runs it under no_grad exactly like
the reference (:141-145), and keeps it out of the optimizer. Without a `cmodel` the encoder consumes precomputed SSL
features [B, 1024, T'] directly.
"""
import torch
import torch.nn as nn

from . import ops
from .modules import ConvParams


class WN(nn.Module):
    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        assert kernel_size % 2 == 1
        if gin_channels != 0:
            raise NotImplementedError('global conditioning (gin_channels) is never used by SSLEncoder (model/ssl_encoder.py:138)')
        if p_dropout != 0:
            raise NotImplementedError('dropout in WN is 0 in SSLEncoder')
        self.hidden_channels, self.kernel_size, self.dilation_rate, self.n_layers = hidden_channels, kernel_size, dilation_rate, n_layers
        self.in_layers, self.res_skip_layers = nn.ModuleList(), nn.ModuleList()
        for i in range(n_layers):
            d = dilation_rate ** i
            self.in_layers.append(ConvParams(hidden_channels, 2 * hidden_channels, kernel_size, pad=(kernel_size * d - d) // 2, dil=d))
            self.res_skip_layers.append(ConvParams(hidden_channels, 2 * hidden_channels if i < n_layers - 1 else hidden_channels, 1))

    def forward(self, x, x_mask=1, g=None):
        if g is not None or not (isinstance(x_mask, int) and x_mask == 1):
            raise NotImplementedError('WN runs unmasked and unconditioned in SSLEncoder (model/ssl_encoder.py:106-108)')
        return ops.wn_stack(x, [m.spec for m in self.in_layers], [m.spec for m in self.res_skip_layers])


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0):
        super().__init__()
        self.out_channels = out_channels
        self.pre = ConvParams(in_channels, hidden_channels, 1, wn=False)
        self.enc = WN(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=gin_channels)
        self.proj = ConvParams(hidden_channels, out_channels * 2, 1, wn=False)

    def forward(self, x):
        """Returns m (the mean half of `proj`). The reference also draws z = m + randn * exp(logs) (:114) and discards
        it in SSLEncoder.forward (:146-148); it is not computed here."""
        x = self.pre(x.contiguous().float())
        x = self.enc(x)
        stats = self.proj(x)
        return stats[:, :self.out_channels].contiguous()


class SSLEncoder(nn.Module):
    def __init__(self, encoder_model='wavlm', num_layers=16, emb_dim=128, kernel_size=5, dilation_rate=1, weight_norm=None, cmodel=None):
        super().__init__()
        if encoder_model != 'wavlm':
            raise NotImplementedError('Unknown encoder model')
        self.encoder_model = encoder_model
        self.ssl_dim = 1024
        if cmodel is not None:
            self.cmodel = cmodel.eval()           # frozen feature extractor: a submodule like in the reference (state_dict keys cmodel.*)
        else:
            self.cmodel = None
        self.encoder = Encoder(self.ssl_dim, emb_dim, emb_dim, kernel_size, dilation_rate, num_layers)
        self._top = None

    def features(self, x):
        """Waveform [B,1,T] -> SSL features [B,1024,T/320] through the injected extractor (model/ssl_encoder.py:141-145)."""
        if self.cmodel is None:
            raise RuntimeError('SSLEncoder was built without a feature extractor (cmodel): pass SSL features [B,1024,T\'] instead of a waveform')
        with torch.no_grad():
            x = torch.nn.functional.pad(x, (160, 0))
            c = self.cmodel.extract_features(x.squeeze(1))[0]
            return c.transpose(1, 2).contiguous()

    def forward(self, x):
        if self._top is not None:
            self._top().begin_forward(x)
        c = x if (x.dim() == 3 and x.shape[1] == self.ssl_dim) else self.features(x)
        return self.encoder(c)
