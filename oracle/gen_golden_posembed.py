"""Golden for the absolute position embedding table (THIS CONTAINER ONLY; imports the reference).

Run:  PYTHONPATH=oracle/shims:/root/reference:. MODEL_DIR=/tmp/mdl python oracle/gen_golden_posembed.py
`get_1d_sincos_pos_embed_from_grid(embed_dim, arange(crop))` (latent_model.py:22-40), as the model's `pos_embed`
buffer is initialised (:151-153), for crop = 9 at embed_dim 384 and 48."""
import os
import numpy as np
from mdgen.model.latent_model import get_1d_sincos_pos_embed_from_grid

out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pos_embed.npz")
np.savez_compressed(out, c384=get_1d_sincos_pos_embed_from_grid(384, np.arange(9)).astype(np.float32),
                    c48=get_1d_sincos_pos_embed_from_grid(48, np.arange(9)).astype(np.float32))
print("wrote", os.path.abspath(out), os.path.getsize(out))
