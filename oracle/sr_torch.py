"""Oracle of the build-defined SR network: the same module as kmsr_amd.sr.SRNet expressed with stock torch ops
(nn.Conv2d / F.relu / nn.PixelShuffle) -- TEST INFRASTRUCTURE ONLY.  The reference ships no SR model (SURVEY.md
section 0.2), so this file, not the reference, is what the HIP conv stack is compared with: "parity unpinned by the
reference".  state_dict keys are identical to SRNet's, so weights interchange."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class ResBlock(nn.Module):
    def __init__(self, n_feats):
        super().__init__()
        self.conv1 = nn.Conv2d(n_feats, n_feats, 3, padding=1)
        self.conv2 = nn.Conv2d(n_feats, n_feats, 3, padding=1)

    def forward(self, x):
        return x + self.conv2(F.relu(self.conv1(x)))


class SRNetTorch(nn.Module):
    def __init__(self, n_colors=3, n_feats=64, n_resblocks=16, scale=4):
        super().__init__()
        self.head = nn.Conv2d(n_colors, n_feats, 3, padding=1)
        self.body = nn.ModuleList([ResBlock(n_feats) for _ in range(n_resblocks)])
        self.body_tail = nn.Conv2d(n_feats, n_feats, 3, padding=1)
        self.up = nn.ModuleList([nn.Conv2d(n_feats, 4 * n_feats, 3, padding=1) for _ in range(int(math.log2(scale)))])
        self.tail = nn.Conv2d(n_feats, n_colors, 3, padding=1)
        self.shuffle = nn.PixelShuffle(2)

    def forward(self, x):
        f = self.head(x)
        h = f
        for blk in self.body:
            h = blk(h)
        h = self.body_tail(h) + f
        for up in self.up:
            h = self.shuffle(up(h))
        return self.tail(h)
