"""GPU: option train_streams 1 vs 2 -- same kernels, same summation orders: the gradients must be bit-identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mdgen_amd.config import ModelConfig
from mdgen_amd.synthetic import synth_state_dict, synth_forward_inputs
from mdgen_amd.train import TrainableModel
dev = torch.device("cuda")
for (B, T, L, tps) in ((2, 40, 72, False), (1, 250, 256, False), (2, 24, 8, True)):
    cfg = ModelConfig.atlas(num_frames=T, crop=L) if not tps else ModelConfig(crop=L, num_frames=T, num_layers=2, tps_condition=True, abs_pos_emb=True)
    sd = synth_state_dict(cfg, 6)
    inp = synth_forward_inputs(cfg, B, T, L, 3, 27)
    gen = torch.Generator().manual_seed(5)
    ut = torch.randn(B, T, L, cfg.latent_dim, generator=gen)
    lm = (torch.rand(B, T, L, cfg.latent_dim, generator=gen) > 0.1).float() * inp["mask"][..., None]
    for prec in (16, 32):
        res = {}
        for ns in (1, 2):
            tm = TrainableModel(cfg, dev).load_state_dict(sd)
            tm.model.set_option("train_precision", prec)
            tm.model.set_option("train_streams", ns)
            kw = {}
            if tps:
                kw["end_frames"] = (inp["end_rot"].to(dev), inp["end_trans"].to(dev))
            args = (inp["x"].to(dev), inp["t"].to(dev), ut.to(dev), lm.to(dev), inp["mask"].to(dev),
                    (inp["start_rot"].to(dev), inp["start_trans"].to(dev)), inp["x_cond"].to(dev), inp["x_cond_mask"].to(dev),
                    inp["aatype"].to(dev))
            for _ in range(3):
                tm.zero_grad()
                loss, _ = tm.forward_backward(*args, **kw)
            torch.cuda.synchronize()
            res[ns] = (float(loss.mean()), tm.grads.clone())
            del tm
        same = torch.equal(res[1][1], res[2][1])
        print(f"B{B} T{T} L{L} tps {tps} precision {prec}: loss {res[1][0]:.6f} / {res[2][0]:.6f}  grads identical: {same}  "
              f"|g| {float(res[1][1].norm()):.4f}  max diff {float((res[1][1] - res[2][1]).abs().max()):.3e}")
        assert same
print("ok")
