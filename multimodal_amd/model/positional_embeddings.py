"""Rotary embedding tables (megatron/model/positional_embeddings.py:38-63).  The table is built with the reference's
own arithmetic - including its quirk that, once the engine has cast the module, `inv_freq` and the positions are in
the module dtype (bf16 positions above 256 are rounded) - and handed to the HIP kernel as an input."""
import torch

from .. import _streams


class RotaryEmbedding(torch.nn.Module):
    def __init__(self, dim, base=10000, precision=torch.half):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)
        self.dim = dim
        self.seq_len_cached = None
        self.cos_cached = None
        self.sin_cached = None
        self.precision = precision
        self._stamp = None  # event behind the kernels that filled the cache (multimodal_amd/_streams.py)

    def forward(self, x, seq_dim=1, seq_len=None):
        """Returns (cos, sin) as [seq_len, 1, 1, dim] like the reference."""
        if seq_len is None:
            seq_len = x.shape[seq_dim]
        if seq_len != self.seq_len_cached or self.cos_cached is None or self.cos_cached.device != x.device:
            self.seq_len_cached = seq_len
            t = torch.arange(seq_len, device=x.device).type_as(self.inv_freq)
            freqs = torch.einsum("i,j->ij", t, self.inv_freq.to(x.device))
            emb = torch.cat((freqs, freqs), dim=-1).to(x.device)
            if self.precision == torch.bfloat16:
                emb = emb.float()
            self.cos_cached = emb.cos()[:, None, None, :]
            self.sin_cached = emb.sin()[:, None, None, :]
            if self.precision == torch.bfloat16:
                self.cos_cached = self.cos_cached.bfloat16()
                self.sin_cached = self.sin_cached.bfloat16()
            self._stamp = _streams.stamp(x.device) if x.is_cuda else None
        elif self._stamp is not None:
            _streams.wait(self._stamp, x.device)
        return self.cos_cached, self.sin_cached

    def tables(self, x, seq_len):
        """[seq_len, dim] contiguous tables in x.dtype for the kernel."""
        cos, sin = self.forward(x, seq_len=seq_len)
        return cos.reshape(seq_len, self.dim).to(x.dtype).contiguous(), sin.reshape(seq_len, self.dim).to(x.dtype).contiguous()
