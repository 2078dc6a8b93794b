"""GPU: the training step's attention kernels alone (mdgen_debug_train_attention, bf16 operands) at the ATLAS per-GPU shape,
both axes, N repetitions -- to be run under rocprofv3 --stats (scripts/r04/attn16_variants.sh), product or experiment library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mdgen_amd import _lib as L
dev = torch.device("cuda")
T, Lr = 250, 256
ntok = T * Lr
gen = torch.Generator().manual_seed(3)
qkv = torch.randn(ntok, 1152, generator=gen)
qkv[:, :384] *= 24 ** -0.5 * 2.0
mask = (torch.rand(ntok, generator=gen) > 0.05).float()
d = lambda t: t.to(dev).contiguous()
g = dict(qkv=d(qkv), mask=d(mask), bk=d(torch.randn(384, generator=gen)), bv=d(torch.randn(384, generator=gen)),
         f=d(1.0 / (10000.0 ** (torch.arange(0, 24, 2).float() / 24))), dout=d(torch.randn(ntok, 384, generator=gen)))
out = torch.empty(ntok, 384, device=dev); lse = torch.empty(ntok, 16, device=dev); dqkv = torch.empty(ntok, 1152, device=dev)
dbias = torch.empty(256, 768, device=dev); stats = torch.empty(ntok, 16, 2, device=dev)
s = L.stream_ptr()
PREC = int(sys.argv[2]) if len(sys.argv) > 2 else 16   # 16: the training step's dispatch; 160: the chunked kernels for every length
axes = {"residue": (T, Lr, T, 0, Lr, 1), "temporal": (Lr, T, Lr, T * Lr, 1, Lr)}   # (nseq, len, inner, outer_stride, inner_stride, pos_stride)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    for name, ax in axes.items():
        L.check(L.lib.mdgen_debug_train_attention(PREC, L.ptr(g["qkv"]), ntok, *ax, L.ptr(g["mask"]), L.ptr(g["bk"]), L.ptr(g["bv"]),
                                                  L.ptr(g["f"]), L.ptr(g["dout"]), L.ptr(out), L.ptr(lse), L.ptr(dqkv), L.ptr(dbias),
                                                  L.ptr(stats), s))
torch.cuda.synchronize()
print("ok", float(out.float().abs().mean()))
