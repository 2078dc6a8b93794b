"""GPU: time one training step (forward + backward [+ Adam]) at cfg-5's per-GPU size (B1 T250 L256, 5 layers) or a given shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdgen_amd.config import ModelConfig
from mdgen_amd.optim import Adam
from mdgen_amd.synthetic import synth_state_dict, synth_forward_inputs
from mdgen_amd.train import TrainableModel
dev = torch.device("cuda")
B, T, L = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (1, 250, 256))]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
prec = int(sys.argv[5]) if len(sys.argv) > 5 else 32   # train_precision: 32 exact fp32 operands, 16 bf16 operands
cfg = ModelConfig.atlas(num_frames=T, crop=L)
sd = synth_state_dict(cfg, 6)
inp = synth_forward_inputs(cfg, B, T, L, 16 if L >= 64 else 0, 27)
gen = torch.Generator().manual_seed(5)
ut = torch.randn(B, T, L, cfg.latent_dim, generator=gen)
lm = (torch.rand(B, T, L, cfg.latent_dim, generator=gen) > 0.1).float() * inp["mask"][..., None]
tm = TrainableModel(cfg, dev).load_state_dict(sd)
tm.model.set_option("train_precision", prec)
for kv in sys.argv[6:]:   # further library options, name=value
    k, v = kv.split("=")
    tm.model.set_option(k, int(v))
args = (inp["x"].to(dev), inp["t"].to(dev), ut.to(dev), lm.to(dev), inp["mask"].to(dev),
        (inp["start_rot"].to(dev), inp["start_trans"].to(dev)), inp["x_cond"].to(dev), inp["x_cond_mask"].to(dev),
        inp["aatype"].to(dev))
opt = Adam(tm.params, lr=1e-4, grad_clip=1.0)
tm.zero_grad(); tm.forward_backward(*args); torch.cuda.synchronize()
t0 = time.time()
for _ in range(reps):
    tm.zero_grad()
    loss, _ = tm.forward_backward(*args)
    opt.step(tm.grads)
    tm.mark_updated()   # no hand-back: the training kernels read the flat parameter buffer
torch.cuda.synchronize()
dt = (time.time() - t0) / reps
print(f"training step B{B} T{T} L{L} train_precision {prec}: {dt * 1e3:.1f} ms  ({B * T / dt:.0f} frames/s); loss {float(loss):.4f}; "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
