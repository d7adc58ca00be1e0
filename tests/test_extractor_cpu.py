"""CPU: the self-contained trunk pieces of extractor.py against the reference's own classes (goldens from
oracle/make_golden.py) -- pure PyTorch, no HIP involved."""
import os

import numpy as np
import torch

from isehr_amd.synth import synth_rows


def test_soa_block_matches_reference(golden_dir):
    from isehr_amd.extractor import SOABlock
    z = np.load(os.path.join(golden_dir, "extractor_tail.npz"))
    soa = SOABlock(32, 4).eval()
    sd = soa.state_dict()
    keys = sorted(sd.keys())
    assert keys == list(z["soa_keys"])                      # same parameter / buffer names as the reference class
    for i, k in enumerate(keys):
        t = sd[k]
        if t.dtype.is_floating_point and t.numel() > 0:
            vals = synth_rows(70 + i, 0, 1, t.numel())[0].reshape(tuple(t.shape)) / 4.0
            if k.endswith("running_var"):
                vals = np.abs(vals) + 0.5
            sd[k] = torch.from_numpy(vals.astype(np.float32))
    soa.load_state_dict(sd)
    x = torch.from_numpy(synth_rows(90, 0, 2 * 32, 6 * 5).reshape(2, 32, 6, 5))
    with torch.no_grad():
        out = soa(x).numpy()
    assert np.abs(out - z["z_soa"]).max() < 1e-5


def test_trunk_shapes():
    from isehr_amd.extractor import ResNet101SOA
    net = ResNet101SOA(blocks=(1, 1, 2, 1), width=8).eval()
    with torch.no_grad():
        y = net(torch.zeros(2, 3, 64, 96))
    assert y.shape == (2, 256, 2, 3) and net.outputdim == 256
    full = ResNet101SOA()
    n_params = sum(p.numel() for p in full.parameters())
    assert full.outputdim == 2048 and 4.2e7 < n_params < 6.5e7      # ResNet-101 trunk (42.5 M) + two SOA blocks
