"""Edge-Enhancing Module operator: Sobel magnitude on the HIP path (reference models/edge_operator.py:29-49).
Prewitt / Canny / edge_loss of the reference are never called by the hot path and are not provided."""
import torch
import torch.nn as nn

from ._backend import lib as _lib
from ._backend import ops


class Sobel(nn.Module):
    def __init__(self, requires_grad=False):
        super().__init__()
        if requires_grad:
            raise NotImplementedError("Sobel HIP path: fixed (non-trainable) filter, as used by Pix2PixModel")
        # parameter container with the reference's key (filter.weight); the kernel hard-codes the same Gx/Gy taps
        self.filter = nn.Conv2d(1, 2, kernel_size=3, stride=1, padding=0, bias=False)
        gx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]])
        gy = torch.tensor([[1.0, 2.0, 1.0], [0.0, 0.0, 0.0], [-1.0, -2.0, -1.0]])
        self.filter.weight = nn.Parameter(torch.stack([gx, gy]).unsqueeze(1), requires_grad=False)

    def forward(self, img):
        _lib.require_gpu(img)
        if img.dim() != 4 or img.shape[1] != 1:
            raise ValueError("Sobel expects a (B,1,H,W) tensor")
        return ops.sobel(img.detach().contiguous().float())
