"""Descriptor extraction with a GPU-resident tail.

The CNN trunk stays in stock PyTorch-ROCm (north star); what changes against the reference is everything
after the last feature map: `extract_vectors` (src/networks/imageretrievalnet.py:356-386) runs batch size 1
and `extract_ms` (:464-479) copies every scale's descriptor to the CPU (`.cpu()` = a device sync per scale
per image), and the features then travel through pickles to the matcher.  Here

    feature map [B, C, H, W]  --HIP-->  GeM -> L2N -> whiten FC -> L2N          (`mi_desc_tail_device`)
    per scale                 --HIP-->  acc += desc ** msp                       (`mi_desc_ms_accumulate_device`)
    after the scales          --HIP-->  (acc / S) ** (1/msp) / ||.||             (`mi_desc_ms_finish_device`)
    descriptors [B, D]        --HIP-->  appended to the device gallery           (`mi_gallery_append_device`)

so descriptors never leave the GPU between extraction and search.

`ResNet101SOA` is a self-contained definition of the reference's trunk (torchvision is not available in this
image): ResNet-101 bottleneck stages with second-order attention blocks after conv4_x and conv5_x
(src/networks/networks.py:94-211).  No checkpoint ships with the reference, so trunk parity is structural
(random weights); `SOABlock` is checked against the reference's own class on seeded weights.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


class SOABlock(nn.Module):
    """Second-order attention (src/networks/networks.py:94-146): z = v(softmax(f(x)^T g(x) / sqrt(mid)) h(x)) + x."""

    def __init__(self, in_ch, k):
        super().__init__()
        self.in_ch, self.mid_ch = in_ch, in_ch // k
        self.f = nn.Sequential(nn.Conv2d(in_ch, self.mid_ch, 1), nn.BatchNorm2d(self.mid_ch), nn.ReLU())
        self.g = nn.Sequential(nn.Conv2d(in_ch, self.mid_ch, 1), nn.BatchNorm2d(self.mid_ch), nn.ReLU())
        self.h = nn.Conv2d(in_ch, self.mid_ch, 1)
        self.v = nn.Conv2d(self.mid_ch, in_ch, 1)
        nn.init.constant_(self.v.weight, 0.0)       # the reference starts the residual branch at zero
        nn.init.constant_(self.v.bias, 0.0)

    def forward(self, x):
        B, C, H, W = x.shape
        f_x = self.f(x).view(B, self.mid_ch, H * W)
        g_x = self.g(x).view(B, self.mid_ch, H * W)
        h_x = self.h(x).view(B, self.mid_ch, H * W)
        attn = torch.softmax((self.mid_ch ** -0.5) * torch.bmm(f_x.permute(0, 2, 1), g_x), dim=-1)
        z = torch.bmm(attn, h_x.permute(0, 2, 1)).permute(0, 2, 1).reshape(B, self.mid_ch, H, W)
        return self.v(z) + x


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        o = F.relu(self.bn1(self.conv1(x)))
        o = F.relu(self.bn2(self.conv2(o)))
        o = self.bn3(self.conv3(o))
        return F.relu(o + idt)


def _stage(inplanes, planes, blocks, stride):
    layers = [_Bottleneck(inplanes, planes, stride, downsample=True)]
    layers += [_Bottleneck(planes * 4, planes) for _ in range(blocks - 1)]
    return nn.Sequential(*layers)


class ResNet101SOA(nn.Module):
    """Trunk of SOLAR_Global_Retrieval (src/networks/networks.py:149-211): ResNet-101 without avgpool/fc, SOA after
    conv4_x (in 1024, k = 4) and conv5_x (in 2048, k = 2).  `blocks` can be shrunk for tests."""

    def __init__(self, blocks=(3, 4, 23, 3), soa_layers="45", width=64):
        super().__init__()
        w = width
        self.conv1 = nn.Sequential(nn.Conv2d(3, w, 7, 2, 3, bias=False), nn.BatchNorm2d(w))
        self.conv2_x = nn.Sequential(nn.ReLU(), nn.MaxPool2d(3, 2, 1), _stage(w, w, blocks[0], 1))
        self.conv3_x = _stage(w * 4, w * 2, blocks[1], 2)
        self.conv4_x = _stage(w * 8, w * 4, blocks[2], 2)
        self.conv5_x = _stage(w * 16, w * 8, blocks[3], 2)
        self.soa4 = SOABlock(w * 16, 4) if "4" in soa_layers else None
        self.soa5 = SOABlock(w * 32, 2) if "5" in soa_layers else None
        self.outputdim = w * 32

    def forward(self, x):
        x = self.conv4_x(self.conv3_x(self.conv2_x(self.conv1(x))))
        if self.soa4 is not None:
            x = self.soa4(x)
        x = self.conv5_x(x)
        if self.soa5 is not None:
            x = self.soa5(x)
        return x


class DescriptorTail:
    """GeM(p, eps) -> L2N -> [Linear(W, b) -> L2N] on the GPU through the C ABI.  `whiten_w` [C_out, C] and
    `whiten_b` [C_out] are float32 cuda tensors (the state of the reference's nn.Linear) or None."""

    def __init__(self, p=3.0, eps=1e-6, whiten_w=None, whiten_b=None):
        self.p, self.eps = float(p), float(eps)
        self.w = whiten_w.contiguous().float() if whiten_w is not None else None
        self.b = whiten_b.contiguous().float() if whiten_b is not None else None

    @property
    def outputdim(self):
        return None if self.w is None else self.w.shape[0]

    def __call__(self, feat):
        """feat: float32 cuda tensor [B, C, H, W] -> descriptors [B, D] (cuda)."""
        feat = feat.contiguous().float()
        B, C, H, W = feat.shape
        d = C if self.w is None else self.w.shape[0]
        out = torch.empty((B, d), dtype=torch.float32, device=feat.device)
        scratch = torch.empty((B, C), dtype=torch.float32, device=feat.device) if self.w is not None else None
        _lib.desc_tail_device(feat.data_ptr(), B, C, H * W, self.p, self.eps,
                              self.w.data_ptr() if self.w is not None else None,
                              self.b.data_ptr() if self.b is not None else None, d,
                              scratch.data_ptr() if scratch is not None else None, out.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
        return out


def extract_ms_device(trunk, tail, images, ms=(1.0, 2 ** 0.5, 2 ** -0.5), msp=1.0):
    """Multi-scale descriptors of a batch (images [B, 3, H, W] cuda), the math of extract_ms
    (src/networks/imageretrievalnet.py:464-479) with every scale and the final normalisation on the device."""
    stream = torch.cuda.current_stream().cuda_stream
    acc = None
    with torch.no_grad():
        for i, s in enumerate(ms):
            x = images if s == 1 else F.interpolate(images, scale_factor=s, mode="bilinear", align_corners=False)
            desc = tail(trunk(x))
            if acc is None:
                acc = torch.empty_like(desc)
            _lib.desc_ms_accumulate_device(acc.data_ptr(), desc.data_ptr(), desc.numel(), msp, i == 0, stream)
        _lib.desc_ms_finish_device(acc.data_ptr(), acc.shape[0], acc.shape[1], len(ms), msp, stream)
    return acc


def extract_to_gallery(trunk, tail, batches, gallery, ms=(1.0,), msp=1.0):
    """Offline step without a CPU round trip: descriptors of every image batch are appended to the device
    gallery (which normalises them like matching_L2 and builds the MFMA image).  Returns the number of rows."""
    stream = torch.cuda.current_stream().cuda_stream
    n = 0
    for images in batches:
        if len(ms) == 1 and ms[0] == 1:
            with torch.no_grad():                      # extract_ss: the net output as it is (:461-462)
                d = tail(trunk(images))
        else:
            d = extract_ms_device(trunk, tail, images, ms, msp)
        gallery.append_device(d.data_ptr(), d.shape[0], stream)
        n += d.shape[0]
    torch.cuda.synchronize()
    return n
