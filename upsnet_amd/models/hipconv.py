"""Route the nn.Conv2d / nn.ConvTranspose2d layers of the inference graph through the hand-written MFMA convolution kernels
(csrc/conv1x1.hip, conv1x1_pair.hip, conv_wino.hip, conv.hip, conv_bf16.hip) with fused bias / residual / ReLU epilogues.

`conv(module, x, relu=False, residual=None)` computes relu?(module(x) + residual). Weights are packed once per
module (cached; re-packed if the parameter changes).

No silent library fallback: a CUDA tensor that reaches a layer the kernels do not cover (dilated or grouped convolutions,
Cin not a multiple of 32, ...) RAISES, unless UPSNET_ALLOW_LIBRARY_CONV=1, in which case the layer runs on the library
convolution (MIOpen) and is recorded in FALLBACKS. CPU tensors always go through the module itself (torch-CPU convolution +
the same epilogue in torch): that is the host execution used by oracle.forward.forward_cpu and by synthetic.calibrate_statistics,
never a product path on a GPU box.

TRACE (parity tests): set to a list and every convolution launch appends a record of its module(s), inputs, fused epilogue
operands and output -- tests/test_layerwise_gpu.py replays each record in float64 ("per op, on identical inputs").
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops

ENABLED = True
ALLOW_LIBRARY = os.environ.get('UPSNET_ALLOW_LIBRARY_CONV', '0') == '1'
FALLBACKS = []   # (layer description, reason) of every CUDA-tensor call that left the hand-written path
TRACE = None


def _library(m, x, why):
    """Called on the way to the module's own (library) forward."""
    if x.is_cuda:
        FALLBACKS.append((repr(m), why))
        if not ALLOW_LIBRARY:
            raise RuntimeError('upsnet_amd.hipconv: %r on a CUDA tensor %s is not covered by the hand-written kernels (%s); set '
                               'UPSNET_ALLOW_LIBRARY_CONV=1 to run it on the library convolution' % (m, tuple(x.shape), why))


def _trace(kind, **kw):
    if TRACE is not None:
        kw['kind'] = kind
        TRACE.append(kw)


# 3x3 / stride-1 layers with enough 2x2 output tiles to occupy half the chip go through the Winograd F(2x2,3x3) kernel
# (csrc/conv_wino.hip; 1.2-2.2x faster there, tools/bench_winograd.py; same fp32 arithmetic class, ~1e-5 absolute difference).
# Smaller layers stay on the direct form.
WINOGRAD = os.environ.get('UPSNET_WINOGRAD', '1') != '0'
WINOGRAD_MIN_WORKGROUPS = int(os.environ.get('UPSNET_WINOGRAD_MIN_WG', '128'))
WINO_TM64_MIN = int(os.environ.get('UPSNET_WINO_TM64_MIN', '768'))   # 64-tile Winograd workgroups above this many (csrc/conv_wino.hip reads the same)
# r11: the largest of those layers (>= one workgroup of 32 4x4-tiles x 64 channels per CU, last round at least WINO36_MIN_FILL full: FPN P2 / P3,
# the 5-level RPN launch, res2's 3x3) go through the F(4x4,3x3) kernel (csrc/conv_wino36.hip): 2.25 multiplies per output instead of 9
# (F(2x2): 4), 1.4-1.6x faster than F(2x2) there (tools/bench_winograd36.py), slower on small maps. Its rounding error is 3-4x that of
# F(2x2) (tools/winograd_error_cpu.py: <= 0.08 of the layer tolerance tests/test_layerwise_gpu.py allows); UPSNET_WINO36=0 switches it off.
WINO36 = os.environ.get('UPSNET_WINO36', '1') != '0'
WINO36_ROI = os.environ.get('UPSNET_WINO36_ROI', '1') != '0'   # the mask head (pinned kernel choice, ROI batches of 14 x 14) on the F(4x4) form as well.
# r12 measured ONE same-box pair (167.6 vs 169.3 img/s, "inside the spread") and left it off; r13, three interleaved pairs on one box: 168.26 /
# 168.31 / 168.47 -> 169.86 / 170.05 / 169.55 img/s (+ 0.9 %, every pair), and the strict end-to-end margins of the mask tensors do not move
# (mask_probs 0.036 / 0.064, pan_mask_logit 0.144 / 0.266 of the 1e-4 bound at 256x512 / 1024x2048; F(2x2): 0.035 / 0.065, 0.150 / 0.253): on.
WINO36_MIN_FILL = float(os.environ.get('UPSNET_WINO36_MIN_FILL', '0.65'))
# r13: split-K F(4x4) for single maps with fewer workgroups than CUs (csrc/conv_wino36.hip, conv_wino36_f32_kernel<true> + reduce). Built and
# measured (tools/bench_winograd36_splitk.py, 1024x2048): res3 conv2 54.8 -> 50.2 us (x2), res4 conv2 / FPN P4 48.2 -> 45.0 (x4), res5 conv2 and
# P5 slower than the F(2x2) split-K form -- 3-5 us on ten launches (~0.7 % of the image) for 2-3x the rounding error on ten chained backbone
# layers: OFF by default, UPSNET_WINO36_SPLITK=1 routes res3 / res4 conv2 and FPN P4 to it.
WINO36_SPLITK = os.environ.get('UPSNET_WINO36_SPLITK', '0') != '0'
SPLITK = os.environ.get('UPSNET_SPLITK', '1') != '0'
# 1x1 convolutions (stride 1 / 2) with >= CONV1X1_MIN_WG workgroups of 64 pixels x 64 channels go through the lean GEMM kernel
# (csrc/conv1x1.hip); smaller ones (res5's 2048-pixel maps: split-K) and Cout < 32 heads stay on the general kernel.
CONV1X1 = os.environ.get('UPSNET_CONV1X1', '1') != '0'
CONV1X1_MIN_WG = int(os.environ.get('UPSNET_CONV1X1_MIN_WG', '256'))
# Arithmetic of the dense convolutions: 'fp32' (default; exact fp32 products on the fp32 MFMA, the configuration every headline
# number is measured on), 'bf16x3' (bf16 matrix cores, 3-term split, fp32-equivalent to ~1e-5) or 'bf16' (BASELINE.json
# configs[2]: bf16 products, fp32 accumulation). The stem, the deconvolution, the FPN top-down laterals and the deformable
# convolutions always use the fp32 kernel.
PRECISION = os.environ.get('UPSNET_CONV_PRECISION', 'fp32')
# 'bf16' mode: the backbone keeps its activations in bf16 between layers (the kernels read / write 2-byte elements; the fp32
# accumulator + bias + residual + ReLU is rounded once per layer). UPSNET_BF16_ACT=0: fp32 activations as in r07 (A/B runs).
BF16_ACT = os.environ.get('UPSNET_BF16_ACT', '1') != '0'


def act_dtype():
    """dtype the backbone stores its activations in: torch.bfloat16 in the 'bf16' mode (BASELINE configs[2]), else float32."""
    return torch.bfloat16 if (PRECISION == 'bf16' and BF16_ACT) else torch.float32


def _plans(m):
    """Packed-weight cache of ONE module, stored on the module itself. (A process-wide dict keyed by id(module) is unsafe: the id
    -- and with the caching allocator even the weight's address -- of a freed model's layer can be reused by a different layer
    of a later model, and a stale pack of another shape makes the kernel read out of bounds.)"""
    d = m.__dict__.get('_hip_plans')
    if d is None:
        d = m.__dict__['_hip_plans'] = {}
    return d



def _plan(m):
    w = m.weight
    key = (w.data_ptr(), w._version, tuple(w.shape), None if m.bias is None else m.bias._version)
    ent = _plans(m).get('direct')
    if ent is None or ent[0] != key:
        wp, ldw = ops.pack_conv_weight(w.detach())
        ent = (key, wp, ldw)
        _plans(m)['direct'] = ent
    return ent[1], ent[2]


def _conv1x1_plan(m):
    w = m.weight
    key = (w.data_ptr(), w._version, tuple(w.shape), None if m.bias is None else m.bias._version)
    ent = _plans(m).get('c1x1')
    if ent is None or ent[0] != key:
        ent = (key, ops.pack_conv1x1_weight(w.detach()))
        _plans(m)['c1x1'] = ent
    return ent[1]


def _use_conv1x1(m, x, always=False):
    if not (CONV1X1 and tuple(m.kernel_size) == (1, 1) and tuple(m.padding) == (0, 0) and m.stride[0] in (1, 2) and
            m.in_channels % 32 == 0 and m.out_channels >= 32):
        return False
    if always:   # layers whose batch size varies at run time: the kernel (hence the summation order) must not depend on it
        return True
    st = m.stride[0]
    pix = x.shape[0] * ((x.shape[2] - 1) // st + 1) * ((x.shape[3] - 1) // st + 1)
    return -(-pix // 64) * -(-m.out_channels // 64) >= CONV1X1_MIN_WG


# A workgroup of the lean 1x1 kernel (4 waves, one per SIMD) keeps its CU's matrix pipe busy by itself (measured: a lone workgroup
# walks K = 1024 in 16 us = the MFMA time of its 512 MFMAs per wave; two on one CU take twice as long). A launch therefore lasts
# (most workgroups on any CU) x (one K walk): 256 workgroups take one walk, 264 take two -- the 50 x 84 map of UPSNet-101-DCN at
# 800x1333 is 66 x 4 = 264 tiles, and its 1024 -> 256 layer ran 38 us against 22 us for 256 tiles. BALANCE: the tiles beyond the last
# full round (or the whole layer when it is less than one round) run with their K walk split over several workgroups (+ one reduce
# launch), so that they spread over the chip. Power-of-two maps (1024x2048) tile evenly and are left alone.
BALANCE = os.environ.get('UPSNET_CONV1X1_BALANCE', '0') == '1'   # measured r10: 118.98 vs 118.40 img/s on UPSNet-101-DCN 800x1333 -- within noise; off by default
_BAL_OVERHEAD_US = 6.0       # extra launch + reduce pass of a split part
_BAL_STEP_US = 0.49          # one 32-channel K step of one 32 x 32 block: 16 MFMAs x 64 cycles at ~2.1 GHz
_CUS = {}


def _cus(device):
    if device.index not in _CUS:
        _CUS[device.index] = torch.cuda.get_device_properties(device).multi_processor_count
    return _CUS[device.index]


def _c1_bn(m_tiles, cout):
    """output channels per workgroup the launcher picks (csrc/conv1x1.hip, conv1x1_frag_launch)."""
    return 128 if (cout > 64 and m_tiles * -(-cout // 128) >= 512) else 64


def _c1_walk_us(m_tiles, cin, cout):
    """(workgroups, microseconds of one workgroup's K walk) of an unsplit launch over m_tiles 64-pixel tiles."""
    bn = _c1_bn(m_tiles, cout)
    return m_tiles * -(-cout // bn), (cin // 32) * (2 if bn == 128 else 1) * _BAL_STEP_US


def _c1_best_split(wgs, walk_us, nsl, cus):
    """split factor (1..16) minimising the time of `wgs` workgroups whose walk is cut `ks` ways: ceil(wgs ks / cus) / ks walks (+ overhead)."""
    best, best_t = 1, -(-wgs // cus) * walk_us
    for ks in range(2, min(16, nsl // 2) + 1):
        if -(-nsl // ks) * (ks - 1) >= nsl:
            continue
        t = -(-wgs * ks // cus) * (walk_us / ks) + _BAL_OVERHEAD_US
        if t < best_t - 1.0:
            best, best_t = ks, t
    return best, best_t


def _c1_balance(m, x, cus=None):
    """-> (rows of the unsplit main part, split factor of the tail part); (all rows, 1) = one plain launch."""
    st = m.stride[0]
    M = x.shape[0] * ((x.shape[2] - 1) // st + 1) * ((x.shape[3] - 1) // st + 1)
    if not BALANCE or st != 1:
        return M, 1
    cus = cus or _cus(x.device)
    cin, cout, nsl = m.in_channels, m.out_channels, m.in_channels // 32
    m_tiles = -(-M // 64)
    wgs, walk = _c1_walk_us(m_tiles, cin, cout)
    plain = -(-wgs // cus) * walk
    if wgs % cus == 0 or wgs >= 8 * cus:
        return M, 1
    nt = wgs // m_tiles
    main_tiles = (wgs // cus) * cus // nt                      # m-tiles of the whole rounds
    main_t = 0.0
    if main_tiles > 0:                                         # (the main part is tiled on its own: its channel tile may differ)
        mw, mwalk = _c1_walk_us(main_tiles, cin, cout)
        main_t = -(-mw // cus) * mwalk
    tw, twalk = _c1_walk_us(m_tiles - main_tiles, cin, cout)
    ks, tail_t = _c1_best_split(tw, twalk, nsl, cus)
    if ks == 1 or main_t + tail_t + (_BAL_OVERHEAD_US if main_tiles else 0.0) >= plain - 1.0:
        return M, 1
    return main_tiles * 64, ks


def _conv1x1_balanced(m, x, relu, residual):
    """conv1x1_frag of a stride-1 layer as (unsplit main rows) + (split-K tail rows), both writing into one output; rows = pixels in
    NHWC order, so each part is a contiguous [rows, C] slab viewed as a 1 x rows image."""
    rows_main, ks = _c1_balance(m, x)
    wp = _conv1x1_plan(m)
    if ks == 1:
        return ops.conv1x1_frag(x, wp, m.bias, m.out_channels, 1, relu=relu, residual=residual), 'conv1x1'
    x = ops.nhwc(x.float())
    N, C, H, W = x.shape
    M, co = N * H * W, m.out_channels
    out = ops._nhwc_out(N, co, H, W, x.device)
    slab = lambda t, c: t.permute(0, 2, 3, 1).reshape(M, c)
    img = lambda rows, c: rows.view(1, 1, rows.shape[0], c).permute(0, 3, 1, 2)
    xr, orow = slab(x, C), slab(out, co)
    rr = None if residual is None else slab(ops.nhwc(residual.float()), co)
    if rows_main:
        ops.conv1x1_frag(img(xr[:rows_main], C), wp, m.bias, co, 1, relu=relu, residual=None if rr is None else img(rr[:rows_main], co),
                         out=img(orow[:rows_main], co))
    ops.conv1x1_frag(img(xr[rows_main:], C), wp, m.bias, co, 1, relu=relu, residual=None if rr is None else img(rr[rows_main:], co),
                     out=img(orow[rows_main:], co), ksplit=ks)
    return out, 'conv1x1 %s splitk%d' % ('main + tail' if rows_main else 'all', ks)


# r13: the small-tile 1x1 kernel (csrc/conv1x1_ksw.hip: 16 pixels x 64 channels of v_mfma_f32_16x16x4_f32 fragments, K split over the four
# waves of a workgroup, no LDS in the K loop) where the 64-pixel tiles of the lean kernel leave the chip idle: the last round of its
# workgroups is less than KSW_MAX_FILL full (UPSNet-101-DCN at 800x1333: 4200-pixel maps are 66 x 4 = 264 workgroups = two K walks for
# 1.03 walks of work) AND the layer has a long K walk into few channels (Cin >= 2 Cout, Cin >= 256: a bottleneck's conv1, the FPN's P5
# lateral) -- the small tile re-fetches the weights once per 16 pixels, which costs more than it spreads for a conv3 (measured,
# tools/bench_conv1x1_ksw.py: 1024 -> 256 on 50 x 84 36.3 -> 27.0 us, 2048 -> 256 on 25 x 42 34.2 -> 19.5, 2048 -> 512 35.2 -> 28.3,
# 256 -> 1024 30.0 -> 30.4). Power-of-two maps (the headline workload) tile evenly and never take it. Shape-only; never for a pinned choice.
KSW = os.environ.get('UPSNET_CONV1X1_KSW', '1') != '0'
KSW_MAX_FILL = float(os.environ.get('UPSNET_CONV1X1_KSW_MAX_FILL', '0.75'))
KSW_TILE = (16, 64)


def _ksw_plan(m):
    w = m.weight
    key = (w.data_ptr(), w._version, tuple(w.shape), None if m.bias is None else m.bias._version)
    ent = _plans(m).get('ksw')
    if ent is None or ent[0] != key:
        ent = (key, ops.pack_conv1x1_ksw_weight(w.detach()))
        _plans(m)['ksw'] = ent
    return ent[1]


def _use_ksw(m, x, cus=None):
    if not (KSW and PRECISION == 'fp32' and x.dtype == torch.float32 and tuple(m.kernel_size) == (1, 1) and tuple(m.padding) == (0, 0) and
            m.stride[0] in (1, 2) and m.in_channels % 16 == 0 and m.out_channels % 4 == 0 and m.in_channels >= 256 and
            m.in_channels >= 2 * m.out_channels):
        return False
    st = m.stride[0]
    pix = x.shape[0] * ((x.shape[2] - 1) // st + 1) * ((x.shape[3] - 1) // st + 1)
    m_tiles = -(-pix // 64)
    wgs = m_tiles * -(-m.out_channels // _c1_bn(m_tiles, m.out_channels))
    cus = cus or _cus(x.device)
    return wgs < KSW_MAX_FILL * -(-wgs // cus) * cus


# r13: 3x3 / stride 1 / pad 1 layers into <= 32 channels (the 18-channel offset predictors of the deformable bottlenecks) on maps too small
# for the Winograd 32-channel form: the small-tile kernel (csrc/conv1x1_ksw.hip, conv3x3_ksw_f32_kernel) instead of the general kernel
# split 6-8 ways over K + its reduce launch (tools/bench_conv3x3_ksw.py: 256 -> 18 on 50 x 84 21.4 -> 14.4 us, 512 -> 18 on 25 x 42
# 21.8 -> 14.9; 23 + 3 such layers per image of UPSNet-101-DCN at 800x1333). Shape-only.
KSW3 = os.environ.get('UPSNET_CONV3X3_KSW', '1') != '0'


def _ksw3_plan(m):
    w = m.weight
    key = (w.data_ptr(), w._version, tuple(w.shape), None if m.bias is None else m.bias._version)
    ent = _plans(m).get('ksw3')
    if ent is None or ent[0] != key:
        ent = (key, ops.pack_conv3x3_ksw_weight(w.detach()))
        _plans(m)['ksw3'] = ent
    return ent[1]


def _use_ksw3(m, x):
    return (KSW3 and PRECISION == 'fp32' and x.dtype == torch.float32 and tuple(m.kernel_size) == (3, 3) and tuple(m.stride) == (1, 1) and
            tuple(m.padding) == (1, 1) and tuple(m.dilation) == (1, 1) and m.out_channels <= 32 and m.in_channels % 16 == 0)


PAIR = os.environ.get('UPSNET_CONV1X1_PAIR', '1') != '0'
PAIR_MIN_TILES = int(os.environ.get('UPSNET_CONV1X1_PAIR_MIN_TILES', '1024'))


PAIR_RES3 = os.environ.get('UPSNET_CONV1X1_PAIR_RES3', '1') != '0'
PAIR_RES4 = os.environ.get('UPSNET_CONV1X1_PAIR_RES4', '1') != '0'


def use_pair(m3, m1, x, residual):
    """conv3 of one bottleneck (m3, + residual + ReLU) and conv1 of the next (m1, + ReLU) in one launch (csrc/conv1x1_pair.hip)?
    Where the workgroups of the pair kernel (64 pixels x all channels) still fill the chip: the res2 stage (64 -> 256 -> 64 on the
    stride-4 map: HBM-bound, the block output is not read back) and -- r10 -- the res3 stage (128 -> 512 -> 128 on the stride-8 map:
    512 workgroups at 1024x2048; one kernel boundary, one prologue and the re-read of the block output less per block boundary) and the
    res4 stage (256 -> 1024 -> 256 on 32-pixel tiles: 256 workgroups at 1024x2048).
    Not in the bf16 modes (there the layers follow hipconv._use_bf16)."""
    if not (PAIR and CONV1X1 and PRECISION == 'fp32' and supported(m3, x) and residual is not None and isinstance(m1, nn.Conv2d)):
        return False
    ok = lambda m: (tuple(m.kernel_size) == (1, 1) and tuple(m.stride) == (1, 1) and tuple(m.padding) == (0, 0) and m.groups == 1)
    if not (ok(m3) and ok(m1) and m3.out_channels % 128 == 0 and m1.in_channels == m3.out_channels):
        return False
    tiles = x.shape[0] * x.shape[2] * x.shape[3] // 64
    if m3.in_channels == 64 and m1.out_channels == 64:
        return tiles >= PAIR_MIN_TILES
    if PAIR_RES3 and m3.in_channels == 128 and m1.out_channels == 128:
        return tiles >= PAIR_MIN_TILES // 2
    if PAIR_RES4 and m3.in_channels == 256 and m1.out_channels == 256:   # 32-pixel tiles: 256 -> 1024 -> 256 on the stride-16 map, >= one workgroup per CU
        return 2 * tiles >= PAIR_MIN_TILES // 4
    return False


def conv_pair(m3, m1, x, residual):
    """(relu(conv3(x) + residual), relu(conv1(that))) -- see use_pair."""
    out1, out2 = ops.conv1x1_pair(x, residual, _conv1x1_plan(m3), m3.bias, m3.out_channels, _conv1x1_plan(m1), m1.bias, m1.out_channels)
    _trace('conv', module=m3, x=x, out=out1, relu=True, residual=residual, residual_up=False, form='pair(conv3)')
    _trace('conv', module=m1, x=out1, out=out2, relu=True, residual=None, residual_up=False, form='pair(conv1)')
    return out1, out2


SIBLINGS = os.environ.get('UPSNET_CONV1X1_SIBLINGS', '1') != '0'


def use_siblings(ma, mb, x):
    """conv1 (ma, + ReLU) and the projection shortcut (mb) of a stage's first bottleneck read the same map with the same stride: one launch
    over the concatenated output channels (csrc/conv1x1.hip, sibling mode) reads it once and saves a kernel boundary. fp32 mode only."""
    if not (SIBLINGS and CONV1X1 and PRECISION == 'fp32' and isinstance(ma, nn.Conv2d) and isinstance(mb, nn.Conv2d) and supported(ma, x) and
            supported(mb, x) and x.dtype == torch.float32):
        return False
    ok = lambda m: tuple(m.kernel_size) == (1, 1) and tuple(m.padding) == (0, 0) and m.groups == 1 and m.stride[0] in (1, 2)
    # only where BOTH layers would take the lean kernel on their own (_use_conv1x1: enough workgroups): on small maps the separate
    # layers go to the implicit-GEMM split-K path, which is faster there, and the results stay the ones of the unfused model
    if not (_use_conv1x1(ma, x) and _use_conv1x1(mb, x)):
        return False
    return (ok(ma) and ok(mb) and ma.stride == mb.stride and ma.in_channels == mb.in_channels and ma.in_channels % 32 == 0 and
            ma.out_channels % 32 == 0 and (ma.bias is None) == (mb.bias is None))


def conv_siblings(ma, mb, x, relu_a=True, relu_b=False):
    """(relu_a?(ma(x)), relu_b?(mb(x))) in one launch -- see use_siblings. Bit-identical to the two launches."""
    key = tuple((m.weight.data_ptr(), m.weight._version, tuple(m.weight.shape), None if m.bias is None else m.bias._version) for m in (ma, mb))
    ent = _plans(ma).get('siblings')
    if ent is None or ent[0] != key:
        wp = ops.pack_conv1x1_weight(torch.cat([ma.weight.detach(), mb.weight.detach()], 0))
        b = None if ma.bias is None else torch.cat([ma.bias.detach(), mb.bias.detach()], 0).contiguous()
        ent = (key, wp, b)
        _plans(ma)['siblings'] = ent
    ya, yb = ops.conv1x1_siblings(x, ent[1], ent[2], ma.out_channels, mb.out_channels, ma.stride[0], relu_a=relu_a, relu_b=relu_b)
    _trace('conv', module=ma, x=x, out=ya, relu=relu_a, residual=None, residual_up=False, form='conv1x1 siblings(a)')
    _trace('conv', module=mb, x=x, out=yb, relu=relu_b, residual=None, residual_up=False, form='conv1x1 siblings(b)')
    return ya, yb


def supported(m, x):
    return (ENABLED and isinstance(m, nn.Conv2d) and x.is_cuda and
            (x.dtype == torch.float32 or (x.dtype == torch.bfloat16 and PRECISION == 'bf16')) and
            ops.conv_supported(m.in_channels, m.kernel_size[0], m.kernel_size[1], m.groups, m.dilation) and
            m.stride[0] == m.stride[1] and m.padding[0] == m.padding[1] and m.padding_mode == 'zeros')


# the bf16 kernels work on 128 x 128 tiles: layers with fewer workgroups than this (res5, the P5 / P6 maps, narrow heads) would leave
# most CUs idle and stay on the fp32 kernels (exact, and faster there)
BF16_MIN_WG = int(os.environ.get('UPSNET_BF16_MIN_WG', '192'))
BF16_MIN_WG_WREG = int(os.environ.get('UPSNET_BF16_MIN_WG_WREG', '16'))


def _use_bf16(m, xs, always=False):
    if PRECISION == 'fp32' or m.kernel_size[0] * m.kernel_size[1] > 9 or m.out_channels < 64:
        return False
    if always or xs[0].dtype == torch.bfloat16:   # layers whose batch size varies at run time: the choice (hence the rounding) must
        return True                               # not depend on it; bf16 activations: only the bf16 kernels read them
    k, st, pd, dl = m.kernel_size[0], m.stride[0], m.padding[0], m.dilation[0]
    wgs = 0
    for x in xs:
        ho, wo = (x.shape[2] + 2 * pd - (dl * (k - 1) + 1)) // st + 1, (x.shape[3] + 2 * pd - (dl * (k - 1) + 1)) // st + 1
        wgs += -(-(x.shape[0] * ho * wo) // 128)
    # (3x3 / 1 layers with 256-channel blocks run on csrc/conv3x3_wreg_bf16.hip, which has 2-row tiles for small maps: P4 / P5 outputs)
    small_ok = PRECISION == 'bf16' and k == 3 and st == 1 and dl == 1 and m.out_channels % 256 == 0
    return wgs * -(-m.out_channels // 128) >= (BF16_MIN_WG_WREG if small_ok else BF16_MIN_WG)


def _bf16_form():
    """Trace label of a bf16-mode launch: the precision + the kernel instance the library picked (only looked up while tracing)."""
    return PRECISION if TRACE is None else '%s %s' % (PRECISION, ops.last_kernel_form())


def _bf16_plan(m):
    w = m.weight
    key = (w.data_ptr(), w._version, tuple(w.shape), None if m.bias is None else m.bias._version, PRECISION)
    ent = _plans(m).get('bf16')
    if ent is None or ent[0] != key:
        ent = (key,) + ops.pack_conv_weight_bf16(w.detach(), split=(PRECISION == 'bf16x3'))
        _plans(m)['bf16'] = ent
    return ent[1], ent[2], ent[3]


# bf16 mode: an identity bottleneck (three layers + shortcut) as one launch whose intermediates stay in the LDS
# (csrc/bottleneck_bf16.hip). UPSNET_BF16_BLOCK=0: three launches (A/B runs).
BF16_BLOCK = os.environ.get('UPSNET_BF16_BLOCK', '1') != '0'
BF16_BLOCK_MIN_TILES = int(os.environ.get('UPSNET_BF16_BLOCK_MIN_TILES', '128'))
BF16_PROJ = os.environ.get('UPSNET_BF16_PROJ', '1') != '0'   # the projection (first) bottleneck of res2-res4 as one launch too


def use_block(blk, x):
    """Can `blk` (a folded, non-deformable bottleneck: identity, or the projection block of res2 / res3 / res4) run as one bf16 launch
    on x?"""
    if not (ENABLED and BF16_BLOCK and PRECISION == 'bf16' and BF16_ACT and x.is_cuda and x.dtype == torch.bfloat16 and not blk.deformable):
        return False
    c1, c2, c3 = blk.conv1, blk.conv2, blk.conv3
    cm, cin, st = c1.out_channels, c1.in_channels, c1.stride[0]
    plain = lambda m, k, p, s=1: (tuple(m.kernel_size) == (k, k) and tuple(m.stride) == (s, s) and tuple(m.padding) == (p, p) and
                                  tuple(m.dilation) == (1, 1) and m.groups == 1)
    if not (plain(c2, 3, 1) and plain(c3, 1, 0) and c2.in_channels == cm and c2.out_channels == cm and c3.in_channels == cm and
            c3.out_channels == 4 * cm):
        return False
    ho, wo = (x.shape[2] - 1) // st + 1, (x.shape[3] - 1) // st + 1
    # (the kernel's tiles are 8x16 / 8x8 / 4x8 pixels at widths 64 / 128 / >= 256: a map with fewer tiles than half the CUs -- res5 at
    # 1024x2048, 64 tiles, 9 MB of weights per block -- stays on the separate layers: measured 156 us fused vs 130 us)
    th, tw = {64: (8, 16), 128: (8, 8)}.get(cm, (4, 8))
    if x.shape[0] * -(-ho // th) * -(-wo // tw) < BF16_BLOCK_MIN_TILES or x.shape[0] * ho * wo * 4 * cm >= (1 << 30):
        return False
    if blk.downsample is None:
        return cm in (64, 128, 256, 512) and plain(c1, 1, 0) and cin == 4 * cm
    pr = blk.downsample[0]
    return (BF16_PROJ and isinstance(pr, nn.Conv2d) and len(blk.downsample) == 2 and isinstance(blk.downsample[1], nn.Identity) and
            st in (1, 2) and plain(c1, 1, 0, st) and plain(pr, 1, 0, st) and pr.in_channels == cin and pr.out_channels == 4 * cm and
            (cm, cin) in ((64, 64), (128, 256), (256, 512)))


def block(blk, x):
    """relu(conv3(relu(conv2(relu(conv1(x))))) + shortcut(x)) -- see use_block."""
    ms = (blk.conv1, blk.conv2, blk.conv3) + (() if blk.downsample is None else (blk.downsample[0],))
    key = tuple((m.weight.data_ptr(), m.weight._version, None if m.bias is None else m.bias._version) for m in ms)
    ent = _plans(blk.conv1).get('block16')
    if ent is None or ent[0] != key:
        pack = ops.pack_bottleneck_bf16 if blk.downsample is None else ops.pack_bottleneck_proj_bf16
        ent = (key, pack(*(m.weight for m in ms), *(m.bias for m in ms)))
        _plans(blk.conv1)['block16'] = ent
    if blk.downsample is None:
        y = ops.bottleneck_bf16(x, ent[1])
    else:
        y = ops.bottleneck_proj_bf16(x, ent[1], blk.conv1.stride[0])
    _trace('block', module=blk, x=x, out=y, form='bottleneck_bf16' if blk.downsample is None else 'bottleneck_proj_bf16')
    return y


def _winograd_plan(m, tn32=False):
    w = m.weight
    key = (w.data_ptr(), w._version, tuple(w.shape), None if m.bias is None else m.bias._version)
    name = 'wino32' if tn32 else 'wino'
    ent = _plans(m).get(name)
    if ent is None or ent[0] != key:
        ent = (key,) + ops.pack_winograd_weight(w.detach(), tn32=tn32)
        _plans(m)[name] = ent
    return ent[1], ent[2]


WINO_TAIL_SPLIT = os.environ.get('UPSNET_WINO_TAIL_SPLIT', '1') != '0'   # r10: the mask head's last partial round on half-size workgroups IN the
# same launch (conv_wino16_tail_f32_kernel): 117 -> 107.5 us per layer in the model (graph-timed alone: 127.7 -> 99.5); as a second launch
# (UPSNET_WINO_TAIL_FUSED=0) it was 114.5


def _wino_tail_split(m, x):
    """A batched launch (the mask head: N ROIs of 14 x 14) on the 32-tile x 64-channel form is ceil(N tiles / 32) x Cout / 64
    workgroups, two resident per CU. 100 ROIs = 616 workgroups = 1.2 rounds of 512: the last 104 run one per CU on 104 CUs while 152
    CUs idle, and the layer takes as long as 1.55 full rounds (115 us, 0.57 of the MFMA peak). Returns n_main > 0 if the batch should
    be split: the first n_main images = whole rounds on the 32 x 64 form, the remaining ones on the 32 x 32 form (twice the workgroups,
    half the work each: at most one per CU) -- same bits from both forms, so a ROI's logits still do not depend on the batch."""
    if not WINO_TAIL_SPLIT or x.shape[0] < 2 or m.out_channels % 64:
        return 0
    cus = _cus(x.device)
    n, tiles, nt = x.shape[0], ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2), m.out_channels // 64
    slots = 2 * cus
    wgs = -(-(n * tiles) // 32) * nt
    if wgs <= slots or _wino_tm(m, [x]) != 32:
        return 0
    n_main = ((wgs // slots) * slots // nt) * 32 // tiles
    # the tail as large as still fits one half-size workgroup per CU (100 ROIs: 80 + 20 -- 99.5 us against 102.5 for 83 + 17)
    fit = (cus // (m.out_channels // 32)) * 32 // tiles
    n_main = min(n_main, n - fit) if n - fit > 0 else n_main
    tail = n - n_main
    if n_main <= 0 or tail <= 0 or -(-(tail * tiles) // 32) * (m.out_channels // 32) > cus:
        return 0
    return n_main


WINO_TAIL_FUSED = os.environ.get('UPSNET_WINO_TAIL_FUSED', '1') != '0'   # both forms in ONE launch (conv_wino16_tail_f32_kernel); 0: two launches


def _wino_split_launch(m, x, n_main, relu):
    wp, ldw = _winograd_plan(m)
    if WINO_TAIL_FUSED:
        wp32, ldw32 = _winograd_plan(m, tn32=True)
        return ops.conv2d_winograd_tail(x, wp, ldw, wp32, ldw32, m.bias, m.out_channels, n_main, relu=relu)
    out = ops._nhwc_out(x.shape[0], m.out_channels, x.shape[2], x.shape[3], x.device)
    ops.conv2d_winograd_multi([x[:n_main]], wp, ldw, m.bias, m.out_channels, relu=relu, outs=[out[:n_main]])
    wp32, ldw32 = _winograd_plan(m, tn32=True)
    ops.conv2d_winograd_multi([x[n_main:]], wp32, ldw32, m.bias, m.out_channels, relu=relu, outs=[out[n_main:]], tn32=True)
    return out


def _winograd36_plan(m):
    w = m.weight
    key = (w.data_ptr(), w._version, tuple(w.shape), None if m.bias is None else m.bias._version)
    ent = _plans(m).get('wino36')
    if ent is None or ent[0] != key:
        ent = (key,) + ops.pack_winograd36_weight(w.detach())
        _plans(m)['wino36'] = ent
    return ent[1], ent[2]


def _winograd36_shape_ok(m):
    return (tuple(m.kernel_size) == (3, 3) and tuple(m.stride) == (1, 1) and tuple(m.padding) == (1, 1) and tuple(m.dilation) == (1, 1) and
            m.in_channels % 32 == 0 and m.out_channels % 64 == 0)


def _use_winograd36(m, xs):
    """The F(4x4,3x3) form (one workgroup = 32 tiles of 4 x 4 outputs x 64 channels, one resident per CU) pays when the launch fills the
    chip: at least one workgroup per CU and a last round that is not mostly idle (each of its rounds costs ~1.4x a round of the F(2x2)
    form, which covers half the area). Never for a pinned kernel choice (ROI batches), residual adds, bf16 inputs or Cout % 64 != 0."""
    if not (WINO36 and WINOGRAD and _winograd36_shape_ok(m) and all(x.dtype == torch.float32 for x in xs)):
        return False
    cus = _cus(xs[0].device)
    wgs = sum(-(-(x.shape[0] * ((x.shape[2] + 3) // 4) * ((x.shape[3] + 3) // 4)) // 32) for x in xs) * (m.out_channels // 64)
    if cus // 2 <= wgs < cus and m.in_channels >= 256:
        # (r13) half a round of workgroups with a long channel walk (>= 16 slabs against the ~9.5 us a workgroup costs besides them): still faster
        # than the F(2x2) form -- UPSNet-101-DCN at 800x1333, FPN P3 on the 100 x 168 map: 132 workgroups, 94.9 vs 120.6 us
        # (tools/bench_winograd36_splitk.py); with 128 input channels (res3 conv2 at 1024x2048: 57.6 vs 54.8) it is not
        return True
    return wgs >= cus and wgs >= WINO36_MIN_FILL * (-(-wgs // cus)) * cus


# (r13, measured and left OFF: the five-level RPN launch at 1024x2048 is 1364 workgroups = 5.33 rounds; cutting it into P2 + P3 on F(4x4) (5 whole
# rounds) and P4-P6 on the F(2x2) multi-map launch -- three interleaved same-box pairs 171.09 / 171.05 / 171.13 img/s without, 170.77 / 170.67 /
# 170.43 with, serial 6.33 vs 6.35 ms: the third-full last round costs less than a launch of its own)
WINO36_PREFIX = os.environ.get('UPSNET_WINO36_PREFIX', '0') != '0'


def _wino36_round_prefix(m, xs):
    """How many leading maps of a multi-map F(4x4) launch to keep in it: all of them, unless the launch's last round of workgroups is less than
    half full AND a prefix of the maps makes whole rounds (last round >= 0.95 full) -- then that prefix. Shape-only."""
    n = len(xs)
    if not WINO36_PREFIX or n < 2:
        return n
    cus = _cus(xs[0].device)
    per = [-(-(x.shape[0] * ((x.shape[2] + 3) // 4) * ((x.shape[3] + 3) // 4)) // 32) * (m.out_channels // 64) for x in xs]
    fill = lambda w: w / float(-(-w // cus) * cus)
    last = sum(per) % cus
    if last == 0 or last >= cus // 2:
        return n
    for k in range(n - 1, 0, -1):
        w = sum(per[:k])
        if w >= cus and fill(w) >= 0.95:
            return k
    return n


def _wino36_ksplit(m, x):
    """Split factor of the opt-in split-K F(4x4) form for one map: the split that brings the launch to one workgroup per CU, >= 4 slabs of 16
    channels each; 1 = not this form (enough workgroups for the unsplit form, or fewer than cus / 8: res5 / P5, where F(2x2) split-K wins)."""
    if not (WINO36 and WINOGRAD and _winograd36_shape_ok(m) and x.dtype == torch.float32):
        return 1
    cus = _cus(x.device)
    wgs = -(-(x.shape[0] * ((x.shape[2] + 3) // 4) * ((x.shape[3] + 3) // 4)) // 32) * (m.out_channels // 64)
    if wgs >= cus or wgs < cus // 4 or (wgs >= cus // 2 and m.in_channels >= 256):   # (the last: the unsplit form takes those, _use_winograd36)
        return 1
    ks = min(cus // wgs, (m.in_channels // 16) // 4, 8)
    return ks if ks >= 2 else 1


def _use_winograd(m, xs, always=False):
    if not (WINOGRAD and tuple(m.kernel_size) == (3, 3) and tuple(m.stride) == (1, 1) and tuple(m.padding) == (1, 1) and
            tuple(m.dilation) == (1, 1) and m.in_channels % 16 == 0):
        return False
    if always:
        return True
    return _wino_workgroups(m, xs, _wino_tm(m, xs)) * _wino_ksplit(m, xs) >= WINOGRAD_MIN_WORKGROUPS


def _wino_workgroups(m, xs, tm=64):
    tiles = sum(-(-(x.shape[0] * ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2)) // tm) for x in xs)
    return tiles * (1 if m.out_channels <= 32 else -(-m.out_channels // 64))


def _wino_tm(m, xs):
    """2x2 tiles per workgroup the launcher will pick (csrc/conv_wino.hip, conv_wino16_launch): 64 (one workgroup per CU) above
    768 such workgroups, else 32 (two per CU); always 32 for the 32-channel form (Cout <= 32)."""
    return 64 if m.out_channels > 32 and _wino_workgroups(m, xs, 64) > WINO_TM64_MIN else 32


def _wino_ksplit(m, xs):
    """Split-K factor of the Winograd kernel for one map with fewer than 256 32-tile x 64-channel workgroups (res5 3x3, FPN
    P5): the largest split that stays within the 512 workgroup slots of 256 CUs and keeps >= 4 slabs of 16 channels per
    workgroup."""
    if not SPLITK or len(xs) != 1 or m.out_channels % 4 or _wino_tm(m, xs) == 64:
        return 1
    wgs, slabs = _wino_workgroups(m, xs, 32), m.in_channels // 16
    if wgs >= 256:   # (res4 conv2, 256 workgroups: 46 us unsplit, 53 us split in two + the reduce launch)
        return 1
    k = 1
    while k < 8 and wgs * (k + 1) <= 512 and slabs // (k + 1) >= 4:
        k += 1
    return k


def _ksplit(m, x, ldw):
    """Split-K factor for maps with too few tiles to fill the chip (measured, tools/bench_splitk.py: pays only below
    ~256 workgroups with a long K walk -- res5 3x3, FPN P5). Narrow heads (ldw = 32: the 18-channel offset predictors of the
    deformable bottlenecks, 128-pixel tiles): a 3x3 / 256 -> 18 layer on the 50 x 84 map of UPSNet-101-DCN at 800x1333 is 33
    workgroups walking 72 slabs one after the other (66 us for 2 us of matrix work, 28 such launches per image): split up to 8 ways."""
    if not SPLITK:
        return 1
    k, st, pd = m.kernel_size[0], m.stride[0], m.padding[0]
    pix = x.shape[0] * ((x.shape[2] + 2 * pd - k) // st + 1) * ((x.shape[3] + 2 * pd - k) // st + 1)
    slabs = k * k * m.in_channels // 32
    if ldw == 32:
        blocks, ks = -(-pix // 128), 1
        while ks < 8 and blocks * (ks + 1) <= 384 and slabs // (ks + 1) >= 6:
            ks += 1
        while ks > 1 and -(-slabs // ks) * (ks - 1) >= slabs:
            ks -= 1
        return ks
    if ldw % 64 or m.out_channels % 4:
        return 1
    blocks = -(-pix // 64) * (ldw // 64)
    if blocks <= 128 and slabs >= 32:
        return 4
    if blocks <= 256 and slabs >= 128:
        return 3
    return 1


def conv(m, x, relu=False, residual=None, residual_up=False, winograd=True, pin=False, out_dtype=None):
    """relu?(m(x) + residual) on the hand-written kernels -- see _conv for the arguments. out_dtype=torch.bfloat16 (bf16 mode, a
    layer the bf16 kernels take): the result is stored as bf16; a bf16 `x` on a layer those kernels do not take is widened first."""
    if x.dtype == torch.bfloat16 and not (supported(m, x) and _use_bf16(m, [x])):
        x = x.float()
    if residual is not None and residual.dtype == torch.bfloat16 and not (supported(m, x) and _use_bf16(m, [x], always=pin or winograd == 'always')):
        residual = residual.float()
    y, form = _conv(m, x, relu, residual, residual_up, winograd, pin or winograd == 'always', out_dtype)
    _trace('conv', module=m, x=x, out=y, relu=relu, residual=residual, residual_up=residual_up, form=form)
    return y


def _conv(m, x, relu=False, residual=None, residual_up=False, winograd=True, pin=False, out_dtype=None):
    """residual_up: `residual` is at half resolution and is added through a nearest x2 upsampling (FPN top-down add).
    winograd=False / 'always' pins the direct / the Winograd form; 'f2x2' = the usual choice WITHOUT the F(4x4,3x3) form (3x3 layers upstream
    of a chain of deformable bottlenecks: models/resnet.py, _Block.feeds_deformable); pin=True (implied by 'always') makes every kernel choice
    (bf16 or fp32, lean 1x1 GEMM or general kernel) independent of the batch size, for layers fed by ROI batches whose size
    varies at run time: the logits of a ROI must not depend on how many other ROIs share the launch."""
    if supported(m, x):
        if (not residual_up or tuple(m.kernel_size) == (1, 1)) and _use_bf16(m, [x], always=pin):
            hi, lo, ldw = _bf16_plan(m)
            od = out_dtype if (out_dtype == torch.bfloat16 and lo is None) else torch.float32
            return ops.conv2d_nhwc_bf16_multi([x], hi, lo, ldw, m.bias, m.out_channels, m.kernel_size[0], m.stride[0], m.padding[0],
                                              relu=relu, residuals=None if residual is None else [residual],
                                              residual_up=residual_up, out_dtype=od)[0], _bf16_form()
        if winograd is True and not pin and residual is None and WINO36_SPLITK and _wino36_ksplit(m, x) > 1:
            wp, ldw = _winograd36_plan(m)
            ks = _wino36_ksplit(m, x)
            return ops.conv2d_winograd36_splitk(x, wp, ldw, m.bias, m.out_channels, ks, relu=relu), 'winograd36 splitk%d' % ks
        if winograd is True and not pin and residual is None and _use_winograd36(m, [x]):
            wp, ldw = _winograd36_plan(m)
            return ops.conv2d_winograd36_multi([x], wp, ldw, m.bias, m.out_channels, relu=relu)[0], 'winograd36'
        if (winograd == 'always' and WINO36 and WINO36_ROI and residual is None and x.dtype == torch.float32 and _winograd36_shape_ok(m) and
                x.shape[2] * x.shape[3] <= 1024):
            # ROI batches (the mask head's 14 x 14 maps; opt-in): the F(4x4) kernel whatever the batch size -- 100 ROIs are 200 workgroups, 0.78
            # of one round: 94 us alone against 100-107 on the F(2x2) form with its half-size tail (tools/bench_winograd36.py)
            wp, ldw = _winograd36_plan(m)
            return ops.conv2d_winograd36_multi([x], wp, ldw, m.bias, m.out_channels, relu=relu)[0], 'winograd36 roi'
        if winograd and not residual_up and _use_winograd(m, [x], always=(winograd == 'always')):
            wp, ldw = _winograd_plan(m)
            ks = 1 if winograd == 'always' else _wino_ksplit(m, [x])
            if ks > 1:
                return ops.conv2d_winograd_splitk(x, wp, ldw, m.bias, m.out_channels, ks, relu=relu, residual=residual), 'winograd splitk%d' % ks
            n_main = _wino_tail_split(m, x) if (residual is None and x.dtype == torch.float32) else 0
            if n_main:
                return _wino_split_launch(m, x, n_main, relu), 'winograd tm32 + tail tn32'
            return ops.conv2d_winograd_multi([x], wp, ldw, m.bias, m.out_channels, relu=relu,
                                             residuals=None if residual is None else [residual])[0], 'winograd tm%d' % _wino_tm(m, [x])
        if not pin and residual is None and _use_ksw3(m, x):
            # (reached only where the Winograd 32-channel form was not taken: maps below WINOGRAD_MIN_WORKGROUPS workgroups)
            return ops.conv3x3_ksw(x, _ksw3_plan(m), m.bias, m.out_channels, relu=relu), 'conv3x3 ksw 16x32'
        if not pin and not residual_up and _use_ksw(m, x):
            # (before the lean / general choice: maps below CONV1X1_MIN_WG workgroups -- res5, the P5 lateral -- are the emptiest launches)
            return ops.conv1x1_ksw(x, _ksw_plan(m), m.bias, m.out_channels, KSW_TILE, stride=m.stride[0], relu=relu,
                                   residual=residual), 'conv1x1 ksw %dx%d' % KSW_TILE
        if _use_conv1x1(m, x, always=pin):
            if not pin and not residual_up and m.stride[0] == 1:
                return _conv1x1_balanced(m, x, relu, residual)
            return ops.conv1x1_frag(x, _conv1x1_plan(m), m.bias, m.out_channels, m.stride[0], relu=relu, residual=residual,
                                    residual_up=residual_up), 'conv1x1'
        wp, ldw = _plan(m)
        ks = 1 if (residual_up or pin) else _ksplit(m, x, ldw)
        if ks > 1:
            return ops.conv2d_nhwc_splitk(x, wp, ldw, m.bias, m.out_channels, m.kernel_size[0], m.stride[0], m.padding[0], ks,
                                          relu=relu, residual=residual), 'igemm splitk%d' % ks
        return ops.conv2d_nhwc(x, wp, ldw, m.bias, m.out_channels, m.kernel_size[0], m.stride[0], m.padding[0],
                               relu=relu, residual=residual, residual_up=residual_up), 'igemm'
    _library(m, x, 'hipconv.supported() is false')
    y = m(x)
    if residual is not None:
        y = y + (F.interpolate(residual, scale_factor=2, mode='nearest') if residual_up else residual)
    return (F.relu(y, inplace=True) if relu else y), 'library'


def conv_multi(m, xs, relu=False):
    """The same conv module applied to several feature maps (FPN levels) in ONE launch."""
    xs = list(xs)
    if len(xs) <= 5 and all(supported(m, x) for x in xs):
        if _use_bf16(m, xs):
            hi, lo, ldw = _bf16_plan(m)
            ys, form = ops.conv2d_nhwc_bf16_multi(xs, hi, lo, ldw, m.bias, m.out_channels, m.kernel_size[0], m.stride[0], m.padding[0], relu=relu), _bf16_form()
        elif _use_winograd36(m, xs):
            wp, ldw = _winograd36_plan(m)
            k = _wino36_round_prefix(m, xs)
            if k < len(xs):
                # (r13) the leading maps fill whole rounds of F(4x4) workgroups; the small trailing maps would be a last round that is mostly empty
                # (the five-level RPN launch at 1024x2048: 1024 + 256 | + 64 + 16 + 4 workgroups = 5 rounds | + a third of a sixth): they go where
                # maps of their size go anyway (the F(2x2) multi-map launch)
                ys = ops.conv2d_winograd36_multi(xs[:k], wp, ldw, m.bias, m.out_channels, relu=relu)
                for x, y in zip(xs[:k], ys):
                    _trace('conv', module=m, x=x, out=y, relu=relu, residual=None, residual_up=False, form='winograd36 multi')
                return ys + conv_multi(m, xs[k:], relu=relu)
            ys, form = ops.conv2d_winograd36_multi(xs, wp, ldw, m.bias, m.out_channels, relu=relu), 'winograd36 multi'
        elif _use_winograd(m, xs):
            wp, ldw = _winograd_plan(m)
            ys, form = ops.conv2d_winograd_multi(xs, wp, ldw, m.bias, m.out_channels, relu=relu), 'winograd tm%d multi' % _wino_tm(m, xs)
        else:
            wp, ldw = _plan(m)
            ys, form = ops.conv2d_nhwc_multi(xs, wp, ldw, m.bias, m.out_channels, m.kernel_size[0], m.stride[0], m.padding[0], relu=relu), 'igemm multi'
        for x, y in zip(xs, ys):
            _trace('conv', module=m, x=x, out=y, relu=relu, residual=None, residual_up=False, form=form)
        return ys
    return [conv(m, x, relu=relu) for x in xs]


def stem_supported(m, x):
    return (ENABLED and isinstance(m, nn.Conv2d) and x.is_cuda and x.dtype == torch.float32 and m.in_channels <= 4 and
            m.kernel_size[1] <= 8 and m.groups == 1 and tuple(m.dilation) == (1, 1) and m.stride[0] == m.stride[1] and
            m.padding[0] == m.padding[1] and m.padding_mode == 'zeros')


def conv_stem(m, x, relu=False):
    """The 7x7/2 stem (Cin = 3) on the MFMA kernel: x is the fp32 NCHW blob or already a [N,4,H,W] channels_last image."""
    if not stem_supported(m, x):
        _library(m, x, 'hipconv.stem_supported() is false')
        y = m(x[:, :m.in_channels] if x.shape[1] != m.in_channels else x)
        return F.relu(y, inplace=True) if relu else y
    w = m.weight
    key = (w.data_ptr(), w._version, tuple(w.shape), None if m.bias is None else m.bias._version)
    ent = _plans(m).get('stem')
    if ent is None or ent[0] != key:
        ent = (key,) + ops.pack_stem_weight(w.detach())
        _plans(m)['stem'] = ent
    is_nhwc4 = x.shape[1] == 4 and x.is_contiguous(memory_format=torch.channels_last)
    x4 = x if is_nhwc4 else ops.image_to_nhwc4(x)
    y = ops.conv2d_stem(x4, ent[1], ent[2], m.bias, m.out_channels, m.kernel_size[0], m.kernel_size[1], m.stride[0], m.padding[0], relu=relu)
    _trace('conv', module=m, x=x4[:, :m.in_channels], out=y, relu=relu, residual=None, residual_up=False, form='stem')
    return y


# Stem convolution + ReLU + max-pool as one launch: csrc/stem_pool.hip (fp32 MFMA; the headline path) / csrc/stem_pool_bf16.hip (bf16
# mode). UPSNET_STEM_POOL=0 / UPSNET_BF16_STEM=0: stem kernel + library max-pool (A/B runs).
STEM_POOL = os.environ.get('UPSNET_STEM_POOL', '1') != '0'
BF16_STEM = os.environ.get('UPSNET_BF16_STEM', '1') != '0'


def _stem_geometry(m, x):
    return (ENABLED and stem_supported(m, x) and m.out_channels == 64 and m.in_channels <= 3 and tuple(m.kernel_size) == (7, 7) and
            tuple(m.stride) == (2, 2) and tuple(m.padding) == (3, 3))


def use_stem_pool(m, x):
    """'bf16' / 'f32' = the fused stem + pool kernel the stem runs on, None = separate launches."""
    if not _stem_geometry(m, x):
        return None
    if PRECISION == 'bf16' and BF16_ACT and BF16_STEM:
        return 'bf16'
    return 'f32' if STEM_POOL else None


def stem_pool(m, x):
    """max_pool2d(relu(m(x)), 3, 2, 1) of the 7x7/2 stem in one launch -- see use_stem_pool. bf16 mode: returns bf16."""
    kind = use_stem_pool(m, x)
    w = m.weight
    key = (w.data_ptr(), w._version, tuple(w.shape), None if m.bias is None else m.bias._version)
    ent = _plans(m).get('stem_pool_' + kind)
    if ent is None or ent[0] != key:
        ent = (key, ops.pack_stem_pool_weight_bf16(w.detach()) if kind == 'bf16' else ops.pack_stem_pool_weight_f32(w.detach()))
        _plans(m)['stem_pool_' + kind] = ent
    is_nhwc4 = x.shape[1] == 4 and x.is_contiguous(memory_format=torch.channels_last)
    x4 = x if is_nhwc4 else ops.image_to_nhwc4(x)
    y = ops.stem_pool_bf16(x4, ent[1], m.bias) if kind == 'bf16' else ops.stem_pool_f32(x4, ent[1], m.bias)
    _trace('stem_pool', module=m, x=x4[:, :m.in_channels], out=y, form='stem + pool bf16' if kind == 'bf16' else 'stem + pool')
    return y


DECONV_FRAG = os.environ.get('UPSNET_DECONV_FRAG', '1') != '0'   # 0: the 2x2 transposed convolution on the general kernel (A/B runs)


def deconv2x2(m, x, relu=False):
    """nn.ConvTranspose2d(k=2, s=2, p=0) (+ ReLU) as one MFMA GEMM with a scatter epilogue."""
    geom = (ENABLED and isinstance(m, nn.ConvTranspose2d) and x.is_cuda and tuple(m.kernel_size) == (2, 2) and
            tuple(m.stride) == (2, 2) and tuple(m.padding) == (0, 0) and tuple(m.output_padding) == (0, 0) and m.groups == 1 and
            tuple(m.dilation) == (1, 1) and m.in_channels % 32 == 0)
    if geom and x.dtype == torch.bfloat16 and PRECISION == 'bf16' and m.in_channels % 64 == 0 and m.out_channels % 32 == 0:
        w = m.weight       # bf16 mode, bf16 activations (the mask head): bf16 MFMA GEMM with the same scatter epilogue
        key = (w.data_ptr(), w._version, tuple(w.shape), None if m.bias is None else m.bias._version)
        ent = _plans(m).get('deconv16')
        if ent is None or ent[0] != key:
            ent = (key,) + ops.pack_deconv2x2_weight_bf16(w.detach())
            _plans(m)['deconv16'] = ent
        y = ops.deconv2x2_bf16(x, ent[1], ent[2], m.bias, m.out_channels, relu=relu)
        _trace('deconv', module=m, x=x, out=y, relu=relu, form='deconv2x2 bf16')
        return y
    if x.dtype == torch.bfloat16:
        x = x.float()
    ok = geom and x.dtype == torch.float32
    if not ok:
        _library(m, x, 'not a 2x2 / stride-2 transposed convolution with Cin % 32 == 0')
        y = m(x)
        return F.relu(y, inplace=True) if relu else y
    w = m.weight
    key = (w.data_ptr(), w._version, tuple(w.shape), None if m.bias is None else m.bias._version)
    if DECONV_FRAG and m.out_channels % 32 == 0:
        # the lean 1x1 GEMM kernel with a scatter epilogue (csrc/conv1x1.hip MODE 2): A fragments as one ds_read_b128 per 8 MFMAs, B from
        # L2 -- the general kernel (csrc/conv.hip) reads two LDS words per MFMA
        ent = _plans(m).get('deconv_frag')
        if ent is None or ent[0] != key:
            ent = (key,) + ops.pack_deconv2x2_weight_frag(w.detach(), m.bias)
            _plans(m)['deconv_frag'] = ent
        y = ops.deconv2x2_frag(x, ent[1], ent[2], m.out_channels, relu=relu)
        _trace('deconv', module=m, x=x, out=y, relu=relu, form='deconv2x2 frag')
        return y
    ent = _plans(m).get('deconv')
    if ent is None or ent[0] != key:
        ent = (key,) + ops.pack_deconv2x2_weight(w.detach())
        _plans(m)['deconv'] = ent
    y = ops.deconv2x2(x, ent[1], ent[2], m.bias, m.out_channels, relu=relu)
    _trace('deconv', module=m, x=x, out=y, relu=relu, form='deconv2x2')
    return y


def dcn(m, x, offset, relu=False):
    """relu?(DeformConv m(x, offset)) for the folded inference graph: the fused kernel applies the ReLU in its epilogue (the module's
    own forward -- DeformConvFunction, the reference's signature -- has no such argument and would cost a separate elementwise launch
    per deformable bottleneck: 30 per image of UPSNet-101-DCN). Same bits: max(v, 0) of the same v."""
    if (ENABLED and x.is_cuda and x.shape[0] == 1 and not torch.is_grad_enabled() and
            ops.fused_dcn_supported(m.in_channels, m.out_channels, m.deformable_groups, m.groups, m.padding, m.stride, m.dilation)):
        return ops.deform_conv_fused([x], [offset], ops.cached_dcn_pack(m.weight), m.bias, m.in_channels, m.out_channels, m.kernel_size,
                                     m.stride, m.padding, m.dilation, relu=relu)[0]
    y = m(x, offset)
    return torch.relu_(y) if relu else y


def clear_cache(model=None):
    """Drop the packed weights of every layer of `model` (they are rebuilt on next use)."""
    if model is not None:
        for m in model.modules():
            m.__dict__.pop('_hip_plans', None)


def conv_multi_cat(ms, xs, return_flat=False):
    """Several sibling convolutions of the same geometry (the RPN's objectness and box-delta 1x1 heads) on the same maps as ONE
    launch over the concatenated output channels: the input maps are read once instead of once per head. Returns one list of
    per-map outputs per module (channel slices of the shared output). Every output channel is computed exactly as in a separate
    launch (same kernel instance, same K order)."""
    xs = list(xs)
    m0 = ms[0]
    same = all(isinstance(m, nn.Conv2d) and m.kernel_size == m0.kernel_size and m.stride == m0.stride and m.padding == m0.padding and
               m.dilation == m0.dilation and m.in_channels == m0.in_channels and (m.bias is None) == (m0.bias is None) for m in ms)
    couts = [m.out_channels for m in ms]
    # (the instance is chosen by ldw: the concatenation must stay on the one the separate heads would use)
    # (bf16 modes: heads narrower than 64 channels stay on the fp32 kernel -- _use_bf16 -- and are concatenated all the same)
    if (not same or any(_use_bf16(m, xs) for m in ms) or xs[0].dtype != torch.float32 or len(xs) > 5 or
            not all(supported(m0, x) for x in xs) or
            (sum(couts) + 31) // 32 != 1 or _use_winograd(m0, xs)):
        res = [conv_multi(m, xs) for m in ms]
        return (res, None, None) if return_flat else res
    key = tuple((m.weight.data_ptr(), m.weight._version, tuple(m.weight.shape), None if m.bias is None else m.bias._version) for m in ms)
    ent = _plans(m0).get('cat')
    if ent is None or ent[0] != key:
        w = torch.cat([m.weight.detach() for m in ms], 0)
        b = None if m0.bias is None else torch.cat([m.bias.detach() for m in ms], 0)
        wp, ldw = ops.pack_conv_weight(w)
        ent = (key, wp, ldw, b)
        _plans(m0)['cat'] = ent
    _, wp, ldw, b = ent
    outs = ops.conv2d_nhwc_multi(xs, wp, ldw, b, sum(couts), m0.kernel_size[0], m0.stride[0], m0.padding[0], relu=False)
    res, c0 = [], 0
    for m, c in zip(ms, couts):
        res.append([o[:, c0:c0 + c] for o in outs])
        for x, y in zip(xs, res[-1]):
            _trace('conv', module=m, x=x, out=y, relu=False, residual=None, residual_up=False, form='igemm multi cat')
        c0 += c
    if return_flat:
        return res, getattr(outs[0], '_ups_flat', None), outs
    return res

