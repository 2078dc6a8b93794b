"""GPU forensic: decode differing K-fragment elements, compare each run with a torch fp32 reference."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict
from mdgen_amd.model import LatentMDGenModel
from mdgen_amd.rigid_utils import Rotation
torch.set_grad_enabled(False)
dev = torch.device("cuda")
cfg = ModelConfig(num_layers=1, crop=4, num_frames=1000)
sd = synth_state_dict(cfg, 0)
m = LatentMDGenModel(cfg); m.load_state_dict(sd)
B, T, L = 16, 1000, 4
gen = torch.Generator().manual_seed(9)
x = torch.randn(B, T, L, 21, generator=gen).to(dev)
t = torch.full((B,), 0.3, device=dev)
mask = torch.ones(B, T, L, device=dev)
q = torch.randn(B, L, 4, generator=gen); q = q / q.norm(dim=-1, keepdim=True)
R = Rotation(quats=q.to(dev)).get_rot_mats()
tr_ = torch.cumsum(2.2 * torch.randn(B, L, 3, generator=gen), 1).to(dev)
cm = torch.zeros(B, T, L, dtype=torch.long, device=dev); cm[:, 0] = 1
xc = torch.zeros(B, T, L, 21, device=dev)
aat = torch.randint(0, 20, (B, L), generator=gen).to(dev)
kw = dict(t=t, mask=mask, start_frames=(R, tr_), x_cond=xc, x_cond_mask=cm, aatype=aat)
lay = m.workspace_layout(B, T, L, 1, False)
os.environ["MDGEN_DEBUG_SKIP"] = "5"
snaps = []
for i in range(4):
    out, tr = m.forward(x, return_trace=True, **kw)
    torch.cuda.synchronize()
    ws = m._ws[(B, T, L, 1, 0)]
    snaps.append(ws[lay.kf:lay.vf].clone())
    h0 = tr["h0"].clone()
    mod = ws[lay.mod:lay.silu_t].view(torch.float32).clone()
C = 384
modrow = mod.numel() // B
mod = mod.view(B, modrow)
sh, sc = mod[:, 3 * C:4 * C], mod[:, 4 * C:5 * C]     # trunk layer 0: shift_t, scale_t
y = torch.nn.functional.layer_norm(h0, (C,), eps=1e-6) * (1 + sc[:, None, None]) + sh[:, None, None]
y = y.to(torch.bfloat16).float()
Wk = sd["layers.0.mha_t.attn.k_proj.weight"].to(dev).to(torch.bfloat16).float()
bk = sd["layers.0.mha_t.attn.k_proj.bias"].to(dev)
k = y @ Wk.T + bk                                      # (B,T,L,384)
kh = k.view(B, T, L, 16, 24)
inv = sd["layers.0.mha_t.attn.rot_emb.inv_freq"].to(dev)
ang = torch.arange(T, device=dev).float()[:, None] * torch.cat([inv, inv])[None]     # (T,24)
cos, sin = ang.cos()[None, :, None, None], ang.sin()[None, :, None, None]
rot = torch.cat([-kh[..., 12:], kh[..., :12]], -1)
kr = (kh * cos + rot * sin)                            # (B,T,L,16,24)
ntile = T // 32 + 1
a = snaps[0].view(torch.bfloat16)
allruns = [s.view(torch.bfloat16).float() for s in snaps]
diffmask = torch.zeros_like(allruns[0], dtype=torch.bool)
for i in range(1, 4):
    diffmask |= allruns[i] != allruns[0]
idx = diffmask.nonzero().flatten()
print("differing bf16 elements over 4 runs:", len(idx))
def decode(e):
    byte = e * 2
    frag = byte // 1536; off = byte % 1536
    tile = frag % ntile; sh_ = frag // ntile; head = sh_ % 16; seq = sh_ // 16
    if off < 1024:
        lane, j, ks = off // 16, (off % 16) // 2, 0
    else:
        o = off - 1024; lane, j, ks = o // 8, (o % 8) // 2, 1
    return seq, head, tile, ks, lane, j
import collections
cnt = collections.Counter()
shown = 0
for e in idx.tolist():
    seq, head, tile, ks, lane, j = decode(e)
    cnt[(head % 4, ks, j, lane >> 4)] += 1
    if shown < 4:
        b, l = seq // L, seq % L
        pos = tile * 32 + (lane & 31); hh = lane >> 5; ee = ks * 8 + j
        feat = 6 * hh + (ee >> 1) + 12 * (ee & 1)
        ref = float(kr[b, pos, l, head, feat]) if pos < T else float("nan")
        unrot = float(kh[b, pos, l, head, feat]) if pos < T else float("nan")
        vals = [float(r[e]) for r in allruns]
        print(f"seq{seq} head{head} tile{tile} ks{ks} lane{lane} j{j} pos{pos} feat{feat}: runs {vals} ref {ref:.4f} unrot {unrot:.4f}")
        shown += 1
# full 12-slot view of a few corrupted (seq, head, tile, lane) groups
seen = set()
for e in idx.tolist():
    seq, head, tile, ks, lane, j = decode(e)
    key = (seq, head, tile, lane)
    if key in seen or len(seen) >= 10 or lane % 16 != 0:
        continue
    seen.add(key)
    base = ((seq * 16 + head) * ntile + tile) * 768          # bf16 index of the fragment
    def slots(run):
        v0 = run[base + lane * 8: base + lane * 8 + 8]
        v1 = run[base + 512 + lane * 4: base + 512 + lane * 4 + 4]
        return torch.cat([v0, v1]).tolist()
    b, l = seq // L, seq % L
    pos = tile * 32 + (lane & 31); hh = lane >> 5
    ref = [float(kr[b, pos, l, head, 6 * hh + (ee >> 1) + 12 * (ee & 1)]) for ee in range(12)]
    bad = [r for r in range(4) if any(abs(a_ - b_) > 0.02 for a_, b_ in zip(slots(allruns[r]), ref))]
    print(f"--- seq{seq} head{head} tile{tile} lane{lane} pos{pos}: bad runs {bad}")
    print("   ref :", [f"{v:7.3f}" for v in ref])
    for r in range(4):
        print(f"   run{r}:", [f"{v:7.3f}" for v in slots(allruns[r])])
print("histogram (head%4, ks, j, quarter-wave):", sorted(cnt.items(), key=lambda kv: -kv[1])[:20])
os.environ["MDGEN_DEBUG_SKIP"] = "0"
