"""`UPSNET_*` environment knobs: which ones exist, which ones are set.

The tuning / diagnosis switches of this package are read from the environment at import or first use (models/hipconv.py, ops.py,
models/resnet_upsnet.py, csrc/*.hip via getenv). None of them is set in a default run, so every `UPSNET_*` variable present in the
environment is a deviation from the benchmarked configuration: bench.py prints them into `config.knobs`, and a name that is not in the
registry below (a typo, a knob of an older build) is an error instead of a silently different kernel mix."""
import os
import re

_ROOT = os.path.dirname(os.path.abspath(__file__))
_NAME = re.compile(r'UPSNET_[A-Z0-9_]+')

# The registry: name -> (default, reader, what). It is the authority for check() -- an installed layout without the .hip / .cpp sources
# knows the same knobs as the source tree -- and tests/test_host_logic_cpu.py asserts that it equals a scan of the sources
# (scan_sources), so a knob added to a kernel file without a line here fails the CPU suite.
REGISTRY = {
    'UPSNET_ALLOW_LIBRARY_CONV': ('0', 'models/hipconv.py', 'a convolution no hand-written kernel covers runs on the library instead of raising'),
    'UPSNET_BF16_ACT': ('1', 'models/hipconv.py', 'bf16 mode: bf16 activations between the layers (0: fp32 activations)'),
    'UPSNET_BF16_BLOCK': ('1', 'models/hipconv.py', 'bf16 mode: one-launch bottlenecks (0: three launches)'),
    'UPSNET_BF16_BLOCK_MIN_TILES': ('128', 'models/hipconv.py', 'bf16 mode: fewest tiles for the one-launch bottleneck'),
    'UPSNET_BF16_HALO': ('1', 'csrc/conv_bf16.hip', 'bf16 3x3 layers on the haloed-patch kernel (0: general kernel)'),
    'UPSNET_BF16_MIN_WG': ('192', 'models/hipconv.py', 'bf16 mode: fewest workgroups for a bf16 layer'),
    'UPSNET_BF16_MIN_WG_WREG': ('16', 'models/hipconv.py', 'bf16 mode: fewest workgroups for the weights-from-L2 3x3 kernel'),
    'UPSNET_BF16_NARROW_BELOW': ('384', 'csrc/conv_bf16.hip', 'bf16 general kernel: 128 x 64 tiles below this many workgroups'),
    'UPSNET_BF16_PROJ': ('1', 'models/hipconv.py', 'bf16 mode: projection bottlenecks as one launch'),
    'UPSNET_BF16_STEM': ('1', 'models/hipconv.py', 'bf16 mode: fused stem + pool kernel'),
    'UPSNET_BF16_WREG': ('1', 'csrc/conv3x3_wreg_bf16.hip', 'bf16 3x3 layers on the weights-from-L2 kernel'),
    'UPSNET_BF16_WREG1': ('1', 'csrc/conv1x1_wreg_bf16.hip', 'bf16 1x1 layers on the no-LDS kernel where it wins'),
    'UPSNET_BF16_WREG_TH': ('auto', 'csrc/conv3x3_wreg_bf16.hip', 'tile rows of the weights-from-L2 3x3 kernel'),
    'UPSNET_CONV1X1': ('1', 'models/hipconv.py', '1x1 layers on the lean GEMM kernel (0: implicit-GEMM kernel)'),
    'UPSNET_CONV1X1_BALANCE': ('0', 'models/hipconv.py', 'split-K tail for unevenly tiled 1x1 layers'),
    'UPSNET_CONV1X1_MIN_WG': ('256', 'models/hipconv.py', 'fewest workgroups for the lean 1x1 kernel'),
    'UPSNET_CONV1X1_KSW': ('1', 'models/hipconv.py', 'small-tile (16x16x4 fragment) 1x1 kernel for maps the 64-pixel tiles do not fill evenly'),
    'UPSNET_CONV1X1_KSW_MAX_FILL': ('0.75', 'models/hipconv.py', 'use it when the last round of 64-pixel workgroups is less than this full'),
    'UPSNET_CONV3X3_KSW': ('1', 'models/hipconv.py', 'small-tile 3x3 kernel for <= 32-channel layers (offset predictors) on small maps'),
    'UPSNET_CONV1X1_PAIR': ('1', 'models/hipconv.py', 'conv3 + next conv1 in one launch'),
    'UPSNET_CONV1X1_PAIR32_WAVES': ('8', 'ops.py', 'waves per workgroup of the res4 pair kernel'),
    'UPSNET_CONV1X1_PAIR_MIN_TILES': ('1024', 'models/hipconv.py', 'fewest tiles for the res2 pair kernel'),
    'UPSNET_CONV1X1_PAIR_RES3': ('1', 'models/hipconv.py', 'pair kernel on res3'),
    'UPSNET_CONV1X1_PAIR_RES4': ('1', 'models/hipconv.py', 'pair kernel on res4'),
    'UPSNET_CONV1X1_SIBLINGS': ('1', 'models/hipconv.py', 'conv1 + projection shortcut in one launch'),
    'UPSNET_CONV_PRECISION': ('fp32', 'models/hipconv.py', 'fp32 | bf16 | bf16x3 (bench.py --conv-precision)'),
    'UPSNET_DCN_BF16': ('1', 'ops.py', 'bf16 mode: fused deformable convolution on the bf16 cores'),
    'UPSNET_DCN_KERNEL': ('frag', 'ops.py', 'generation of the fused deformable kernel'),
    'UPSNET_DCN_SMALL_GRID': ('0', 'csrc/deform_fused.hip', 'small-grid variant threshold of the fused deformable kernel'),
    'UPSNET_DCN_SPLITK': ('1', 'ops.py', 'split-K for the offset predictors of small maps'),
    'UPSNET_DCN_VARIANT': ('5', 'csrc/deform_fused.hip', 'schedule variant of the fused deformable kernel'),
    'UPSNET_DECONV_FRAG': ('1', 'models/hipconv.py', 'transposed convolution on the lean GEMM kernel'),
    'UPSNET_DIST_BACKEND': ('nccl', 'upsnet_end2end_test.py', 'torch.distributed backend of the harness (gloo when ranks share a GPU)'),
    'UPSNET_EARLY_MASK': ('1', 'models/resnet_upsnet.py', 'mask head on the side stream as soon as the detections exist'),
    'UPSNET_FC_RELU': ('1', 'models/rcnn.py', 'ReLU of fc6 / fc7 in the library GEMM epilogue (0: separate elementwise launch)'),
    'UPSNET_GRAPH': ('1', 'models/resnet_upsnet.py', 'whole forward as one HIP graph'),
    'UPSNET_GRAPH_ALIAS': ('1', 'models/resnet_upsnet.py', 'outputs are views into the graph buffers (0: copies)'),
    'UPSNET_GRAPH_OWN_STREAM': ('0', 'models/resnet_upsnet.py', 'capture / replay on an own stream with one instance'),
    'UPSNET_GRAPH_SLOTS': ('2', 'models/resnet_upsnet.py', 'graph instances per input shape'),
    'UPSNET_HIP_MEMSET': ('0', 'csrc/fill.hip', 'zero scratch with hipMemsetAsync instead of a kernel'),
    'UPSNET_LIB_PATH': ('', '_lib.py', 'path of libupsnet_hip.so (same-box A/B of two builds)'),
    'UPSNET_NMS_LDS': ('0', 'csrc/nms.hip', 'stage the suppression mask in LDS'),
    'UPSNET_NMS_SCAN16': ('1', 'csrc/nms.hip', 'unrolled scan for <= 1024 boxes'),
    'UPSNET_OVERLAP': ('1', 'models/resnet_upsnet.py', 'semantic branch on a side stream'),
    'UPSNET_PIN': ('1', 'upsnet_end2end_test.py', 'pin each rank to its CPU slice'),
    'UPSNET_ROI_KERNEL': ('auto', 'csrc/roi_align.hip', 'ROIAlign kernel variant 0..4; auto = 3 (corner-sharing) for >= 100 bins per ROI, else 0 -- so '
                                                        '"0" is NOT the default; upsnet_roi_tuning(v >= 0) overrides the variable, (v < 0) returns to it'),
    'UPSNET_ROI_XCD_ORDER': ('1', 'ops.py', 'deal the ROIs of the box head\'s ROIAlign launch to the XCDs by image neighbourhood (table written by prop_merge_kernel)'),
    'UPSNET_ROI_XCD_ORDER_MIN': ('512', 'ops.py', 'only for launches with at least this many ROIs'),
    'UPSNET_ROI_XCD_ORDER_STANDALONE': ('0', 'ops.py', 'build the table with a launch of its own for ROI sets that do not come from pyramid_proposals'),
    'UPSNET_ROI_PER_BIN': ('0', 'csrc/roi_align.hip', 'one wave per (roi, bin) (older decomposition)'),
    'UPSNET_SHARE_GPU': ('0', 'upsnet_end2end_test.py', 'let N ranks share one GPU (functional runs of the N > 1 path)'),
    'UPSNET_SPLITK': ('1', 'models/hipconv.py', 'split-K forms for small maps'),
    'UPSNET_STEM_POOL': ('1', 'models/hipconv.py', 'fused stem + pool kernel (0: stem kernel + library pool)'),
    'UPSNET_WINO36': ('1', 'models/hipconv.py', 'largest 3x3 / stride 1 layers on the Winograd F(4x4,3x3) kernel'),
    'UPSNET_WINO36_SPLITK': ('0', 'models/hipconv.py', 'res3 / res4 conv2 and FPN P4 on the split-K F(4x4) form (3-5 us per launch; more rounding error)'),
    'UPSNET_WINO36_PREFIX': ('0', 'models/hipconv.py', 'multi-map F(4x4) launches keep only the leading maps that fill whole rounds of workgroups (measured: no gain)'),
    'UPSNET_WINO36_ROI': ('1', 'models/hipconv.py', 'mask head (ROI batches) on the Winograd F(4x4,3x3) kernel (0: F(2x2) with the half-size tail)'),
    'UPSNET_WINO36_MIN_FILL': ('0.65', 'models/hipconv.py', 'F(4x4,3x3) only if its last round of workgroups is at least this full'),
    'UPSNET_WINOGRAD': ('1', 'models/hipconv.py', '3x3 / stride 1 layers on the Winograd kernel'),
    'UPSNET_WINOGRAD_MIN_WG': ('128', 'models/hipconv.py', 'fewest workgroups for the Winograd kernel'),
    'UPSNET_WINO_TAIL_FUSED': ('1', 'models/hipconv.py', 'mask-head tail inside the main launch (0: two launches)'),
    'UPSNET_WINO_TAIL_SPLIT': ('1', 'models/hipconv.py', 'mask-head tail on half-size workgroups'),
    'UPSNET_WINO_TM64_MIN': ('768', 'models/hipconv.py', '64-tile Winograd workgroups above this many'),
}


def known():
    """Every UPSNET_* name the package reads (the registry above)."""
    return set(REGISTRY)


def scan_sources():
    """Every UPSNET_* name some source file of the package (or bench.py) mentions -- for the CPU test that keeps REGISTRY honest."""
    names = set()
    files = [os.path.join(os.path.dirname(_ROOT), 'bench.py')]
    for d, _, fs in os.walk(_ROOT):
        files += [os.path.join(d, f) for f in fs if f.endswith(('.py', '.hip', '.cpp', '.h')) and f != 'knobs.py']
    for f in files:
        try:
            with open(f, errors='ignore') as fh:
                names.update(_NAME.findall(fh.read()))
        except OSError:
            pass
    return names


def active(env=None):
    """{name: value} of the UPSNET_* variables set in the environment (= the non-default knobs of this run)."""
    env = os.environ if env is None else env
    return {k: env[k] for k in sorted(env) if k.startswith('UPSNET_')}


def check(env=None):
    """active(), after making sure every set knob is one the sources read; raises ValueError on an unknown name."""
    act = active(env)
    unknown = sorted(set(act) - known())
    if unknown:
        raise ValueError('unknown UPSNET_* environment variable(s) %s -- no source file reads them (typo?)' % ', '.join(unknown))
    return act
