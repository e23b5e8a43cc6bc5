"""GPU: the rows either side of the trunk that were still on library kernels -- input blob kernel, the 7x7/2 stem
and the 2x2 deconvolution on the MFMA convolution kernel."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MEANS = np.array((102.9801, 115.9465, 122.7717,))


@pytest.mark.parametrize("H,W,target,max_size", [(64, 128, 64, 128), (50, 75, 50, 100), (48, 80, 40, 60), (37, 53, 64, 96), (60, 45, 30, 1000)])
@pytest.mark.parametrize("nhwc4", [True, False])
def test_prep_image_u8_vs_oracle(H, W, target, max_size, nhwc4):
    """Bit-exact vs the C oracle for identity, down- and up-scaling, both blob layouts."""
    import oracle
    from upsnet_amd import ops
    rng = np.random.default_rng(H * W)
    im = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    ref, scale = oracle.prep_image(im, MEANS, target, max_size)
    s2, resized, padded = oracle.blob_geometry(H, W, target, max_size)
    assert s2 == scale
    out = ops.prep_image_u8(torch.from_numpy(im).cuda(), MEANS, scale, resized, padded, nhwc4=nhwc4)
    if nhwc4:
        assert out.shape == (1, 4, padded[0], padded[1]) and out.is_contiguous(memory_format=torch.channels_last)
        assert float(out[:, 3].abs().max()) == 0.0
        out = out[:, :3]
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_prep_image_u8_golden_from_reference_python():
    from upsnet_amd.dataset.blob import get_image_blob
    from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
    update_config_dict(CITYSCAPES_R50)
    g = np.load(os.path.join(GOLD, "input_blob.npz"))
    for tag in ("even", "ragged"):
        target, max_size = [int(v) for v in g[tag + "_cfg"]]
        b = get_image_blob(g[tag + "_im"], target, max_size, MEANS, nhwc4=False)
        np.testing.assert_array_equal(b['data'].cpu().numpy(), g[tag + "_blob"])
        assert b['im_info'].tolist() == [[g[tag + "_im"].shape[0], g[tag + "_im"].shape[1], 1.0]]


@pytest.mark.parametrize("N,H,W,Cout,k,stride,pad", [(1, 64, 96, 64, 7, 2, 3), (2, 33, 47, 64, 7, 2, 3), (1, 40, 40, 32, 3, 1, 1), (1, 31, 45, 96, 5, 2, 2)])
def test_stem_conv_vs_torch(N, H, W, Cout, k, stride, pad):
    """conv (Cin = 3) + bias + ReLU through the 4-channel NHWC loader vs torch fp64 conv2d, 1e-4."""
    from upsnet_amd import ops
    torch.manual_seed(H + W)
    x = torch.randn(N, 3, H, W, device='cuda') * 50
    w = torch.randn(Cout, 3, k, k, device='cuda') / (3 * k * k) ** 0.5
    b = torch.randn(Cout, device='cuda')
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad).clamp_min(0).float()
    wp, ldw = ops.pack_stem_weight(w)
    out = ops.conv2d_stem(ops.image_to_nhwc4(x), wp, ldw, b, Cout, k, k, stride, pad, relu=True)
    assert out.shape == ref.shape
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-3)  # |x| ~ 50: abs tol scaled


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(3, 14, 14, 256, 256), (1, 5, 9, 64, 32), (2, 33, 40, 32, 80)])
def test_deconv2x2_vs_torch(N, H, W, Cin, Cout):
    from upsnet_amd import ops
    torch.manual_seed(N + H)
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cin, Cout, 2, 2, device='cuda') / Cin ** 0.5
    b = torch.randn(Cout, device='cuda')
    ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2).clamp_min(0).float()
    wp, ldw = ops.pack_deconv2x2_weight(w)
    out = ops.deconv2x2(x, wp, ldw, b, Cout, relu=True)
    assert out.shape == ref.shape == (N, Cout, 2 * H, 2 * W)
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_model_from_uint8_image_equals_model_from_fp32_blob():
    """uint8 image -> device input kernel -> NHWC4 -> stem == the reference-shaped fp32 NCHW blob -> same stem: identical results."""
    from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
    update_config_dict(CITYSCAPES_R50)
    from upsnet_amd.dataset.blob import get_image_blob
    from upsnet_amd.synthetic import build_model
    model = build_model()
    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, size=(250, 500, 3), dtype=np.uint8)   # padded to 256 x 512
    with torch.no_grad():
        a = get_image_blob(im, 250, 500, MEANS, nhwc4=True)
        b = get_image_blob(im, 250, 500, MEANS, nhwc4=False)
        assert a['data'].shape == (1, 4, 256, 512) and b['data'].shape == (1, 3, 256, 512)
        oa = model({'data': a['data'], 'im_info': a['im_info']})
        ob = model({'data': b['data'], 'im_info': b['im_info']})
    for k in ('panoptic_outputs', 'pred_boxes', 'cls_probs', 'panoptic_cls_inds'):
        assert torch.equal(oa[k], ob[k]), k
