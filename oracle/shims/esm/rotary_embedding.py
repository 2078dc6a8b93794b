"""Stand-in for fair-esm's esm.rotary_embedding (not installed here).

Restates the published rotary embedding: inv_freq = 1/10000^(arange(0,dim,2)/dim); angle table
over the KEY length; x*cos + rotate_half(x)*sin with rotate_half(x) = cat(-x[d/2:], x[:d/2]);
q uses the first q_len rows. Call sites in the reference: mdgen/model/mha.py:13,130,356-357.
"""
import torch


def rotate_half(x):
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


def apply_rotary_pos_emb(x, cos, sin):
    cos = cos[:, : x.shape[-2], :]
    sin = sin[:, : x.shape[-2], :]
    return (x * cos) + (rotate_half(x) * sin)


class RotaryEmbedding(torch.nn.Module):
    def __init__(self, dim: int, *_, **__):
        super().__init__()
        inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)

    def _tables(self, x, seq_dimension=-2):
        seq_len = x.shape[seq_dimension]
        t = torch.arange(seq_len, device=x.device).type_as(self.inv_freq)
        freqs = torch.einsum("i,j->ij", t, self.inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1).to(x.device)
        return emb.cos()[None, :, :], emb.sin()[None, :, :]

    def forward(self, q, k):
        cos, sin = self._tables(k, seq_dimension=-2)
        return apply_rotary_pos_emb(q, cos, sin), apply_rotary_pos_emb(k, cos, sin)
