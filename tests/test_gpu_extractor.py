"""GPU: the extractor tail kernels against the reference's own layer functions (goldens), the appendable device
gallery, and the extract -> gallery -> search pipeline without a CPU round trip."""
import os

import numpy as np
import pytest

import oracle
from isehr_amd.synth import synth_rows

pytestmark = pytest.mark.gpu


def _feats():
    import torch
    B, C = 3, 64
    return [torch.from_numpy(synth_rows(60 + i, 0, B * C, h * w).reshape(B, C, h, w)).cuda()
            for i, (h, w) in enumerate(((5, 7), (7, 10), (4, 5)))]


def _tail(p=3.0, whiten=True):
    import torch
    from isehr_amd.extractor import DescriptorTail
    W = (torch.from_numpy(synth_rows(51, 0, 48, 64)) / 8.0).cuda()
    b = (torch.from_numpy(synth_rows(52, 0, 1, 48)[0]) / 8.0).cuda()
    return DescriptorTail(p, 1e-6, W if whiten else None, b if whiten else None)


def test_tail_single_scale_vs_reference_layers(golden_dir):
    z = np.load(os.path.join(golden_dir, "extractor_tail.npz"))
    f = _feats()
    got = _tail()(f[0]).cpu().numpy()
    assert got.shape == (3, 48) and np.abs(got - z["tail_ss"]).max() < 2e-6
    got2 = _tail(2.5, whiten=False)(f[0]).cpu().numpy()
    assert got2.shape == (3, 64) and np.abs(got2 - z["tail_ss_nowhiten"]).max() < 2e-6


@pytest.mark.parametrize("msp,key", [(1.0, "v_ms1"), (2.0, "v_ms2")])
def test_multiscale_average_vs_reference_extract_ms(golden_dir, msp, key):
    import torch
    from isehr_amd import _lib
    z = np.load(os.path.join(golden_dir, "extractor_tail.npz"))
    tail, f = _tail(), _feats()
    s = torch.cuda.current_stream().cuda_stream
    acc = None
    for i in range(3):
        d = tail(f[i][:1])
        if acc is None:
            acc = torch.empty_like(d)
        _lib.desc_ms_accumulate_device(acc.data_ptr(), d.data_ptr(), d.numel(), msp, i == 0, s)
    _lib.desc_ms_finish_device(acc.data_ptr(), 1, 48, 3, msp, s)
    torch.cuda.synchronize()
    assert np.abs(acc.cpu().numpy()[0] - z[key]).max() < 2e-6


def test_appended_gallery_equals_one_shot_gallery():
    import torch
    from isehr_amd._lib import Gallery
    n, d, nq, k = 5000, 160, 12, 40
    g = synth_rows(3, 0, n, d)
    q = synth_rows(4, 0, nq, d)
    ref = Gallery.from_host(g)
    ri, rs, _ = ref.search(q, k)
    ref.close()
    G = Gallery.empty(6000, d)
    gd = torch.from_numpy(g).cuda()
    s = torch.cuda.current_stream().cuda_stream
    for lo, hi in ((0, 7), (7, 300), (300, 301), (301, 4100), (4100, n)):       # ragged appends across tile borders
        G.append_device(gd[lo:hi].data_ptr(), hi - lo, s)
    torch.cuda.synchronize()
    assert G.n == n
    i1, s1, _ = G.search(q, k)
    assert np.array_equal(i1, ri) and np.array_equal(s1, rs)
    with pytest.raises(RuntimeError):
        G.append_device(gd.data_ptr(), 2000, s)                                 # capacity exceeded
    G.close()


def test_extract_to_gallery_pipeline():
    """Tiny trunk: images -> trunk (PyTorch) -> HIP tail -> appended gallery -> search; against the same pipeline
    evaluated with plain torch ops and the oracle's exact search."""
    import torch
    from isehr_amd._lib import Gallery
    from isehr_amd.extractor import ResNet101SOA, DescriptorTail, extract_to_gallery, extract_ms_device
    torch.manual_seed(0)
    trunk = ResNet101SOA(blocks=(1, 1, 1, 1), width=8).cuda().eval()
    D = trunk.outputdim
    W = (torch.randn(D, D) / D ** 0.5).cuda()
    b = (torch.randn(D) * 0.01).cuda()
    tail = DescriptorTail(3.0, 1e-6, W, b)
    imgs = [torch.randn(4, 3, 64, 64, device="cuda") for _ in range(5)]
    G = Gallery.empty(64, D)
    assert extract_to_gallery(trunk, tail, imgs, G, ms=(1.0, 2 ** 0.5, 2 ** -0.5)) == 20

    def torch_desc(x):
        with torch.no_grad():
            v = 0
            for s in (1.0, 2 ** 0.5, 2 ** -0.5):
                xs = x if s == 1 else torch.nn.functional.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False)
                f = trunk(xs)
                o = f.clamp(min=1e-6).pow(3.0).mean(dim=(2, 3)).pow(1 / 3.0)
                o = o / (o.norm(dim=1, keepdim=True) + 1e-6)
                o = torch.nn.functional.linear(o, W, b)
                v = v + o / (o.norm(dim=1, keepdim=True) + 1e-6)
            v = v / 3
            return (v / v.norm(dim=1, keepdim=True)).cpu().numpy()

    ref_rows = np.concatenate([torch_desc(x) for x in imgs])
    got_rows = G.get_rows(0, 20)
    assert np.abs(got_rows - ref_rows / np.linalg.norm(ref_rows, axis=1, keepdims=True)).max() < 5e-6
    qd = extract_ms_device(trunk, tail, imgs[2][:2], (1.0, 2 ** 0.5, 2 ** -0.5)).cpu().numpy()
    idx, sc, _ = G.search(qd, 5)
    G.close()
    assert idx[0, 0] == 8 and idx[1, 0] == 9                                    # each image retrieves itself
    s = oracle.exact_scores_f64(ref_rows, qd)
    assert oracle.check_topk_parity(idx, s, 5, 2e-5) == []


def test_full_depth_resnet101_soa_to_gallery():
    """BASELINE configs[1] at its real width and depth: the full ResNet101 (3, 4, 23, 3) x 64 trunk with both SOA blocks
    (src/networks/networks.py:149-211; 2048 output channels) under PyTorch-ROCm on the GPU -> HIP descriptor tail with a
    2048 x 2048 whitening layer -> appended 2048-d device gallery -> search.  Random weights (no checkpoint ships): what is
    checked is the pipeline at full size against the same trunk output pushed through plain torch ops."""
    import torch
    from isehr_amd._lib import Gallery
    from isehr_amd.extractor import ResNet101SOA, DescriptorTail, extract_to_gallery
    torch.manual_seed(1)
    trunk = ResNet101SOA().cuda().eval()
    assert trunk.outputdim == 2048 and sum(p.numel() for p in trunk.parameters()) > 42_000_000
    D = 2048
    W = (torch.randn(D, D) / D ** 0.5).cuda()
    b = (torch.randn(D) * 0.01).cuda()
    tail = DescriptorTail(3.0, 1e-6, W, b)
    imgs = torch.randn(3, 3, 224, 224, device="cuda")
    G = Gallery.empty(8, D)
    try:
        assert extract_to_gallery(trunk, tail, [imgs[:2], imgs[2:]], G) == 3
        with torch.no_grad():
            f = trunk(imgs)
            assert f.shape[1] == 2048 and torch.isfinite(f).all()
            o = f.clamp(min=1e-6).pow(3.0).mean(dim=(2, 3)).pow(1 / 3.0)
            o = o / (o.norm(dim=1, keepdim=True) + 1e-6)
            o = torch.nn.functional.linear(o, W, b)
            o = (o / (o.norm(dim=1, keepdim=True) + 1e-6)).cpu().numpy()
        ref = o / np.linalg.norm(o, axis=1, keepdims=True)
        got = G.get_rows(0, 3)
        assert np.abs(got - ref).max() < 5e-6
        idx, sc, _ = G.search(ref, 3)
        assert list(idx[:, 0]) == [0, 1, 2] and np.abs(sc[:, 0] - 1.0).max() < 1e-5
    finally:
        G.close()
