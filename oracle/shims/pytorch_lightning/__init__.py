"""Stand-in for pytorch_lightning: LightningModule == nn.Module + no-op hooks (wrapper.py:46-53)."""
import torch


class LightningModule(torch.nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")


class Trainer:  # never used by the oracle
    def __init__(self, *a, **k):
        raise NotImplementedError
