"""View module (upsnet/operators/modules/view.py)."""
import torch.nn as nn


class View(nn.Module):
    def __init__(self, *shape):
        super(View, self).__init__()
        self.shape = shape

    def forward(self, x):
        return x.view(*self.shape)
