"""UPSNet (ResNet-FPN + RPN + box/mask heads + deformable FCN head + parameter-free panoptic head),
inference branch of upsnet/models/resnet_upsnet.py:88-248 for the MI355X.

`forward(data, label=None)` takes the reference's input dict ({'data': [1,3,H,W] fp32 BGR minus pixel
means, 'im_info': [[H, W, scale]]}) and returns the reference's result dict (cls_probs, pred_boxes,
mask_probs, fcn_outputs, cls_inds, panoptic_cls_inds, panoptic_cls_probs, panoptic_outputs).

Two execution styles produce identical results:
  * pipeline='fused' (default): proposals, detection selection, mask removal and the panoptic fusion
    are device pipelines with fixed-size buffers + device-side counts; the host reads two counters
    once (to size the mask-head batch) and nothing of size [k,H,W] is materialised.
  * pipeline='modules': the reference's module-by-module dataflow (PyramidProposal -> ... ->
    MaskRemoval -> SegTerm -> cat/argmax) through the drop-in modules, materialising what the
    reference materialises. Used by the parity tests.
"""
import gc
import os

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from ..config.config import config
from ..operators.modules.mask_removal import MaskRemoval
from ..operators.modules.mask_roi import MaskROI, check_count
from ..operators.modules.pyramid_proposal import PyramidProposal
from ..operators.modules.unary_logits import SegTerm
from . import hipconv
from .fcn import FCNHead
from .fpn import FPN
from .rcnn import RCNN, MaskBranch
from .resnet import ResNetBackbone, fold_frozen_bn, resnet_rcnn
from .rpn import RPN, rpn_forward_levels


_SIDE = {}  # device index -> (side stream, fork event, join event)
_SLOT_STREAMS = {}  # (device index, graph instance number) -> the stream that instance is captured and replayed on


class _Pending(object):
    """Handle of resnet_upsnet.forward_async()."""

    def __init__(self, model, data, ent, out=None):
        self.model, self.data, self.ent, self.out = model, data, ent, out

    def result(self):
        if self.out is None:
            ent, self.ent = self.ent, None
            ent['done'].synchronize()
            self.out = self.model._forward_fused(self.data, st=ent['out'], counters=ent['host'].tolist())
        return self.out


class resnet_upsnet(resnet_rcnn):

    def __init__(self, backbone_depth, pipeline='fused'):
        super(resnet_upsnet, self).__init__()
        self.pipeline = pipeline
        # semantic head on a side stream, concurrent with the detection chain (a CU-masked side stream that keeps a few CUs
        # free for the latency-bound kernels was measured too: 66-75 img/s vs 80-81, so it is an ordinary stream)
        self.overlap_streams = os.environ.get('UPSNET_OVERLAP', '1') != '0'
        # the static-shape part of the forward (everything before the first host read) replayed as one HIP graph
        self.use_graph = os.environ.get('UPSNET_GRAPH', '1') != '0'
        self._graphs = {}
        # graph instances (each with its own static input / activation / output buffers) used in turn per input shape: with 2,
        # forward_async() can launch image i+1 while the caller still reads the outputs of image i (they stay valid for
        # graph_slots forwards of that shape)
        self.graph_slots = max(1, int(os.environ.get('UPSNET_GRAPH_SLOTS', '2')))
        # (r01-r05 kept a single instance for the linear capture of UPSNET_OVERLAP=0: it faulted on the GPU at replay. Root cause,
        # found by bisection (tools/diag_graph_matrix.sh): the hipMemsetAsync calls of the selection ops became MEMSET NODES of the
        # captured graph, and a linear graph with memset nodes faults on this ROCm stack at replay; the forked capture happened
        # to survive. The ops now zero their scratch with an ordinary kernel (csrc/fill.hip), a captured forward holds kernel
        # nodes only, and every configuration replays -- tests/test_model_gpu.py::test_hip_graph_replay_equals_eager[False].)
        self.graph_outputs_alias = os.environ.get('UPSNET_GRAPH_ALIAS', '1') != '0'
        self.early_mask_head = os.environ.get('UPSNET_EARLY_MASK', '1') != '0'
        self.taps = None  # set to a dict to record the inputs/outputs of every custom-op stage (parity tests)
        self.num_classes = config.dataset.num_classes
        self.num_seg_classes = config.dataset.num_seg_classes
        self.num_reg_classes = (2 if config.network.cls_agnostic_bbox_reg else config.dataset.num_classes)

        self.resnet_backbone = ResNetBackbone(backbone_depth)
        self.fpn = FPN(feature_dim=config.network.fpn_feature_dim, with_norm=config.network.fpn_with_norm,
                       upsample_method=config.network.fpn_upsample_method)
        self.rpn = RPN(num_anchors=config.network.num_anchors, input_dim=config.network.fpn_feature_dim)
        self.rcnn = RCNN(self.num_classes, self.num_reg_classes, dim_in=config.network.fpn_feature_dim,
                         with_norm=config.network.rcnn_with_norm)
        self.mask_branch = MaskBranch(self.num_classes, dim_in=config.network.fpn_feature_dim,
                                      with_norm=config.network.rcnn_with_norm)
        self.fcn_head = FCNHead(config.network.fpn_feature_dim, self.num_seg_classes,
                                num_layers=config.network.fcn_num_layers, with_norm=config.network.fcn_with_norm,
                                upsample_rate=4)
        self.mask_roi = MaskROI(clip_boxes=True, bbox_class_agnostic=False, top_n=config.test.max_det,
                                num_classes=self.num_classes, score_thresh=config.test.score_thresh)
        self.enable_void = config.train.panoptic_box_keep_fraction < 1
        self.mask_roi_panoptic = MaskROI(clip_boxes=True, bbox_class_agnostic=False, top_n=config.test.max_det,
                                         num_classes=self.num_classes, nms_thresh=0.5, class_agnostic=True,
                                         score_thresh=config.test.panoptic_score_thresh)
        self.mask_removal = MaskRemoval(fraction_threshold=0.3)
        self.seg_term = SegTerm(config.dataset.num_seg_classes)
        self.pyramid_proposal = PyramidProposal(
            feat_stride=config.network.rpn_feat_stride, scales=config.network.anchor_scales,
            ratios=config.network.anchor_ratios, rpn_pre_nms_top_n=config.test.rpn_pre_nms_top_n,
            rpn_post_nms_top_n=config.test.rpn_post_nms_top_n, threshold=config.test.rpn_nms_thresh,
            rpn_min_size=config.test.rpn_min_size, individual_proposals=config.train.rpn_individual_proposals)

    # ------------------------------------------------------------------ inference preparation
    def prepare_inference(self, channels_last=True, fold_bn=True):
        """eval(), freeze, optionally fold frozen BN and switch activations/weights to channels-last."""
        self.eval()
        for p in self.parameters():
            p.requires_grad = False
        if fold_bn:
            fold_frozen_bn(self)
        if channels_last:
            self.to(memory_format=torch.channels_last)
        self._channels_last = channels_last
        self.invalidate_graphs()
        return self

    # ------------------------------------------------------------------ shared trunk
    def _pyramid(self, data):
        x = data['data']   # fp32 NCHW blob (the reference's layout) or a [N,4,H,W] channels_last image from the input kernel
        if getattr(self, '_channels_last', False) and x.shape[1] != 4 and not hipconv.stem_supported(self.resnet_backbone.conv1.conv1, x):
            x = x.contiguous(memory_format=torch.channels_last)
        res2, res3, res4, res5 = self.resnet_backbone(x)
        return self.fpn(res2, res3, res4, res5)

    def _trunk(self, data):
        pyramid = self._pyramid(data)
        _, rpn_bbox_pred, rpn_cls_prob = rpn_forward_levels(self.rpn, list(pyramid))
        return pyramid, rpn_cls_prob, rpn_bbox_pred

    def _side_stream(self):
        """(side stream, fork event, join event) of the current device; kept outside the module so that the model stays
        deep-copyable / picklable."""
        dev = torch.cuda.current_device()
        if dev not in _SIDE:
            _SIDE[dev] = (torch.cuda.Stream(), torch.cuda.Event(), torch.cuda.Event())
        return _SIDE[dev]

    def __deepcopy__(self, memo):
        """Deep copies (oracle.forward.cpu_copy) carry the weights, not the captured HIP graphs."""
        import copy
        graphs, self._graphs = self._graphs, {}
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            new.__dict__ = copy.deepcopy(self.__dict__, memo)
        finally:
            self._graphs = graphs
        return new

    def invalidate_graphs(self):
        """Drop the captured HIP graphs (they hold the addresses of the packed weights of capture time). Called by
        load_state_dict / prepare_inference; call it yourself after modifying parameters in place."""
        self._graphs = {}

    def load_state_dict(self, *args, **kwargs):
        self.invalidate_graphs()
        return super().load_state_dict(*args, **kwargs)

    def _class_map_dev(self, dev):
        cm = getattr(self, '_cmap_dev', None)
        if cm is None or cm.device != dev:
            cm = self._cmap_dev = self.seg_term.class_map.to(dev)
        return cm

    def _ev_det(self):
        dev = torch.cuda.current_device()
        if ('det', dev) not in _SIDE:
            _SIDE[('det', dev)] = torch.cuda.Event()
        return _SIDE[('det', dev)]

    def _tap(self, **kw):
        if self.taps is not None:
            self.taps.update({k: (v.detach().clone() if isinstance(v, torch.Tensor) else
                                  [t.detach().clone() for t in v] if isinstance(v, (list, tuple)) else v) for k, v in kw.items()})

    def forward(self, data, label=None):
        if label is not None:
            raise NotImplementedError("upsnet_amd implements the inference branch (label=None) only")
        if self.pipeline == 'modules':
            return self._forward_modules(data)
        return self._forward_fused(data)

    def forward_async(self, data):
        """Launch the forward of one image and return a handle; handle.result() waits for it and returns the output dict of
        forward(). On the graph path the launch is one input copy + one hipGraphLaunch + one small asynchronous device-to-host
        copy, so the caller can launch image i+1 before reading image i (graph_slots >= 2): the host work between two images
        (result read-back, Python, the next launch) then overlaps with the device. Elsewhere it simply runs forward()."""
        x = data['data']
        if (self.pipeline == 'fused' and self.use_graph and self.graph_slots >= 2 and self.taps is None and hipconv.TRACE is None and not ops.PROFILE['enabled']
                and x.is_cuda and not torch.is_grad_enabled()):
            ent = self._phase1_graphed(x, data['im_info'], wait=False)
            if ent is not None and ent['host'] is not None:
                with torch.cuda.stream(ent['stream']):
                    ent['host'].copy_(ent['out']['tail']['counters'], non_blocking=True)
                    ent['done'].record()
                return _Pending(self, data, ent)
            if ent is not None:
                torch.cuda.current_stream().wait_stream(ent['stream'])
            return _Pending(self, data, None, self._forward_fused(data, st=None if ent is None else ent['out'], try_graph=False))
        return _Pending(self, data, None, self.forward(data))

    # ------------------------------------------------------------------ MI355X pipeline
    def _phase1(self, x, im_info, tail=False):
        """Everything up to the first host read (static shapes: fixed-capacity ROI / detection buffers + device counters), on the
        current stream + the side stream; returns device tensors only. Capturable as one HIP graph."""
        pyramid = self._pyramid({'data': x})
        feats = list(pyramid[:4])
        # x4 upsampling of the semantic logits is fused into the panoptic kernel (enable_void branch, rate 4)
        fuse_up = self.enable_void and self.fcn_head.upsample_rate == 4
        # The semantic head (offset convs + deformable convs, ~2 ms of MFMA-bound work) depends only on the FPN outputs; the
        # proposal -> box head -> detection-selection chain (~2 ms, a dozen latency-bound single-workgroup kernels) depends only on
        # the RPN. The RPN convolution (MFMA-bound itself) runs first on its own; then the semantic head is issued on a side
        # stream so that it overlaps with that chain; the main stream joins it at the end of this phase.
        _, rpn_bbox_pred, rpn_cls_prob = rpn_forward_levels(self.rpn, list(pyramid))
        main = torch.cuda.current_stream()
        side, ev_fork, ev_join = self._side_stream() if self.overlap_streams else (main, None, None)
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing and side is not main:
            # events recorded inside a capture belong to that graph: never share them with the eager path
            ev_fork, ev_join = torch.cuda.Event(), torch.cuda.Event()
        if side is not main:
            ev_fork.record(main)
            side.wait_event(ev_fork)
        with torch.cuda.stream(side):
            if fuse_up:
                fcn = self.fcn_head.forward_score(*feats)
            else:
                fcn = self.fcn_head(*feats)['fcn_output']
            if side is not main:
                ev_join.record(side)
                if not capturing:
                    fcn.record_stream(main)

        rois, _, n_rois = self.pyramid_proposal.forward_padded(rpn_cls_prob, rpn_bbox_pred, im_info)
        rcnn_output = self.rcnn(feats, rois, n_rois)
        cls_prob = F.softmax(rcnn_output['cls_score'], dim=1)
        bbox_pred = rcnn_output['bbox_pred']
        self._tap(rpn_cls_prob=rpn_cls_prob, rpn_bbox_pred=rpn_bbox_pred,
                  im_info=im_info, rois=rois, n_rois=n_rois, cls_prob=cls_prob, bbox_pred=bbox_pred)
        # both detection selections are launched back to back; ONE host read of the counters (in forward)
        det_boxes, det_scores, det_cls, det_src, det_num = self.mask_roi.forward_padded(rois, bbox_pred, cls_prob, im_info, n_rois)
        # The mask head of the per-class detections starts as soon as they exist -- on the side stream (idle once the semantic
        # head is done), on the fixed first max_det rows of the padded buffer (rows past the count are zero boxes), i.e. before
        # the host knows the counts: it overlaps with the second selection, the dedup and the host read instead of following them
        max_det = min(int(config.test.max_det), det_boxes.shape[0])
        mask_det = None
        if self.early_mask_head and max_det > 0:
            if side is not main:
                ev_det = torch.cuda.Event() if capturing else self._ev_det()
                ev_det.record(main)
                side.wait_event(ev_det)
            with torch.cuda.stream(side):
                mask_det = self.mask_branch(feats, det_boxes[:max_det])
                if side is not main:
                    ev_join.record(side)   # (re-recorded: the join now also covers the mask head)
                    if not capturing:
                        mask_det.record_stream(main)
        pan_boxes, pan_scores, pan_cls, pan_src, pan_num = self.mask_roi_panoptic.forward_padded(rois, bbox_pred, cls_prob, im_info, n_rois)
        # The two selections overlap heavily (same (ROI, class) -> box table): panoptic detections that are also per-class
        # detections reuse the mask logits computed for those (bit-identical, every ROI goes through the head independently)
        pan_row, extra_boxes, extra_num = ops.mask_roi_dedup(det_src, det_cls, det_num, pan_src, pan_cls, pan_boxes, pan_num)
        nums = None   # (the three counters as one tensor: only the eager continuation needs it; the fixed-capacity tail packs its own)
        if side is not main:
            main.wait_event(ev_join)
        tail_out = None
        if tail and fuse_up and mask_det is not None:
            # The rest of the forward on fixed-capacity buffers + device counters, so that it can live in the same HIP graph:
            # assumes the common case (every panoptic detection is also a per-class detection, <= max_det detections, <= 256
            # panoptic detections); the host checks the counters after the replay and redoes this part eagerly otherwise.
            K = min(256, pan_boxes.shape[0])
            pb, ps, pc = pan_boxes[:K], pan_scores[:K], pan_cls[:K]
            pan_logit = ops.mask_logit_gather(mask_det, pan_row[:K], pc)   # rows / classes clamped inside: one launch
            H, W = fcn.shape[2] * 4, fcn.shape[3] * 4
            keep, num_keep, real_keep = self.mask_removal.select(pb[:, 1:], ps, pan_logit, pc, (H, W), num_dev=pan_num)
            num_stuff = self.num_seg_classes - (self.num_classes - 1)
            panoptic, sem = ops.panoptic_fuse_up(fcn, 4, num_stuff, pb, pan_logit, pc, keep, num_keep, real_keep, self._class_map_dev(pb.device))
            kept_cls, kept_scores, counters = ops.panoptic_tail_pack(keep, num_keep, pc, ps, det_num, pan_num, extra_num)
            tail_out = dict(keep=keep, panoptic=panoptic, sem=sem, counters=counters, mask_prob=torch.sigmoid(mask_det),
                            kept_cls=kept_cls, kept_scores=kept_scores)
        if tail_out is None:
            nums = torch.cat([det_num, pan_num, extra_num])
        return dict(feats=feats, fcn=fcn, fuse_up=fuse_up, tail=tail_out, det_boxes=det_boxes, det_scores=det_scores, det_cls=det_cls,
                    pan_boxes=pan_boxes, pan_scores=pan_scores, pan_cls=pan_cls, pan_row=pan_row, extra_boxes=extra_boxes, nums=nums,
                    mask_det=mask_det, max_det=max_det, _events=(ev_fork, ev_join))

    def _phase1_graphed(self, x, im_info_host, wait=True):
        """HIP-graph replay of _phase1 for this input shape / im_info: the ~150 launches of the trunk, the semantic head (side
        stream) and the proposal / detection chain cost one hipGraphLaunch on the host. graph_slots instances per shape are used
        in turn; for each, the first two images run eagerly (weight packing, function attributes, library handles), the third is
        captured."""
        key = (tuple(x.shape), x.dtype, x.is_contiguous(), tuple(float(v) for v in np.asarray(im_info_host).reshape(-1)))
        slots = self._graphs.get(key)
        if slots is None:
            slots = self._graphs[key] = {'next': 0, 'slots': [{'seen': 0, 'idx': i} for i in range(self.graph_slots)]}
        ent = slots['slots'][slots['next']]
        slots['next'] = (slots['next'] + 1) % len(slots['slots'])
        if 'graph' not in ent:
            ent['seen'] += 1
            if ent['seen'] <= 2:
                return None
            static_x = x.clone()
            static_im = torch.from_numpy(np.asarray(im_info_host, dtype=np.float32).reshape(-1)[:3].copy()).to(x.device)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            sk = None
            if self.graph_slots >= 2 or os.environ.get('UPSNET_GRAPH_OWN_STREAM') == '1':
                # one stream per instance NUMBER, shared by every model / input shape of the process: a stream is bound to one of the
                # few hardware queues when it is created, and a second model with fresh streams can land both of its instances on
                # one queue (bench.py's configs[2] leg: 161 instead of 188 img/s before this)
                sk_key = (x.device.index, ent['idx'])
                if sk_key not in _SLOT_STREAMS:
                    _SLOT_STREAMS[sk_key] = torch.cuda.Stream(device=x.device)
                sk = _SLOT_STREAMS[sk_key]
            gc_was_on = gc.isenabled()
            gc.disable()   # a collection in the middle of the capture could release device objects (illegal while capturing)
            try:
                # (thread_local: other threads of the process -- e.g. the RCCL watchdog of torch.distributed -- may call the runtime
                # while this thread captures)
                with torch.cuda.graph(g, stream=sk, capture_error_mode='thread_local'):
                    out = self._phase1(static_x, static_im, tail=True)
            except Exception as e:   # capture not possible on this stack: stay eager, say so once
                import warnings
                warnings.warn("upsnet_amd: HIP graph capture failed (%s); running eagerly" % (e,))
                self.use_graph = False
                return None
            finally:
                if gc_was_on:
                    gc.enable()
            host = torch.empty((4,), dtype=out['tail']['counters'].dtype).pin_memory() if out.get('tail') is not None else None
            # (the graph reads both static inputs by address)
            ent.update(graph=g, x=static_x, im_info=static_im, out=out, host=host, done=torch.cuda.Event(), stream=sk)
        # Each instance lives on its own stream (captured and replayed there): two images in flight then really overlap on the
        # device -- the kernels of one fill the tail rounds, the small layers and the latency-bound chain of the other (7.6 vs
        # 8.2 ms per image, tools/two_stream_probe.py). Every buffer a graph writes comes from its own capture pool, so
        # concurrent replays of different instances share read-only data only (weights, anchors, class map).
        sk, cur = ent['stream'], torch.cuda.current_stream()
        if sk is None:
            ent['x'].copy_(x)
            ent['graph'].replay()
            return ent
        sk.wait_stream(cur)   # x is ready, and whatever still reads this instance's previous outputs on `cur` is done
        with torch.cuda.stream(sk):
            ent['x'].copy_(x)
            ent['graph'].replay()
        if wait:   # the synchronous forward() continues on the caller's stream
            cur.wait_stream(sk)
        return ent   # (ent['out']: the device tensors the graph writes; no reference from them back to ent -- a cycle would leave
                     # the release of a dropped graph to the garbage collector, which may run in the middle of a later capture)

    def _forward_fused(self, data, st=None, counters=None, try_graph=True):
        """st: the state of an already launched graph replay (forward_async); counters: its four counters, already on the host."""
        x, im_info = data['data'], data['im_info']
        if st is None and try_graph and self.use_graph and self.taps is None and hipconv.TRACE is None and not ops.PROFILE['enabled'] and x.is_cuda:
            ent = self._phase1_graphed(x, im_info)
            st = None if ent is None else ent['out']
        graphed = st is not None
        if st is None:
            st = self._phase1(x, im_info)
        feats, fuse_up = st['feats'], st['fuse_up']
        fcn_score, fcn_output = (st['fcn'], None) if fuse_up else (None, st['fcn'])
        H, W = (fcn_score.shape[2] * 4, fcn_score.shape[3] * 4) if fuse_up else fcn_output.shape[2:]
        t = st.get('tail')
        if t is not None:   # whole forward was in the graph: ONE host read
            n_det, n_pan, n_extra, k = counters if counters is not None else t['counters'].tolist()
            check_count(n_det), check_count(n_pan)
            if n_extra == 0 and n_det <= st['max_det'] and n_pan <= min(256, st['pan_boxes'].shape[0]):
                # every output is a view into the graph's buffers: no launch after the replay. They are valid until this graph
                # instance is replayed again, graph_slots forwards of this shape later (the usual contract of graph-replayed
                # inference); graph_outputs_alias = False (UPSNET_GRAPH_ALIAS=0) returns private copies instead.
                out = {
                    'cls_probs': st['det_scores'][:n_det], 'pred_boxes': st['det_boxes'][:n_det], 'mask_probs': t['mask_prob'][:n_det],
                    'fcn_outputs': t['sem'], 'cls_inds': st['det_cls'][:n_det], 'panoptic_cls_inds': t['kept_cls'][:k],
                    'panoptic_cls_probs': t['kept_scores'][:k], 'panoptic_outputs': t['panoptic'],
                }
                return out if self.graph_outputs_alias else {key: v.clone() for key, v in out.items()}
        else:
            n_det, n_pan, n_extra = st['nums'].tolist()
            check_count(n_det), check_count(n_pan)
        det_boxes, det_scores, det_cls = st['det_boxes'][:n_det], st['det_scores'][:n_det], st['det_cls'][:n_det]
        if graphed:   # results handed to the caller must not alias the graph's static buffers (overwritten by the next replay)
            det_boxes, det_scores, det_cls = det_boxes.clone(), det_scores.clone(), det_cls.clone()
        pan_boxes, pan_scores, pan_cls = st['pan_boxes'][:n_pan], st['pan_scores'][:n_pan], st['pan_cls'][:n_pan]
        pan_row, extra_boxes = st['pan_row'], st['extra_boxes']

        # one mask-head pass over the union of both ROI sets (same weights); the per-class detections' part normally already
        # ran in phase 1 (every ROI goes through the head independently: same bits)
        mask_det = st['mask_det']
        if mask_det is not None and n_det <= st['max_det']:
            mask_score = mask_det[:n_det] if not n_extra else torch.cat([mask_det[:n_det], self.mask_branch(feats, extra_boxes[:n_extra])], 0)
        else:
            mask_score = self.mask_branch(feats, torch.cat([det_boxes, extra_boxes[:n_extra]], 0) if n_extra else det_boxes)
        mask_prob = torch.sigmoid(mask_score[:n_det])
        ms = config.network.mask_size
        pan_logit = mask_score.index_select(0, pan_row[:n_pan].long()).gather(1, pan_cls.view(-1, 1, 1, 1).expand(-1, -1, ms, ms))

        self._tap(det_boxes=det_boxes, det_scores=det_scores, det_cls=det_cls, pan_boxes=pan_boxes, pan_scores=pan_scores,
                  pan_cls=pan_cls, pan_logit=pan_logit)
        keep, num_keep, real_keep = self.mask_removal.select(pan_boxes[:, 1:], pan_scores, pan_logit, pan_cls, (H, W))
        num_stuff = self.num_seg_classes - (self.num_classes - 1)
        cmap = self._class_map_dev(pan_boxes.device)
        if fuse_up:
            self._tap(fcn_score=fcn_score)
        else:
            self._tap(fcn_output=fcn_output)
        if n_pan > 256:  # beyond the fused kernels' instance table: reference-shaped materialising path
            if fcn_output is None:
                fcn_output = F.interpolate(fcn_score, None, 4, mode='bilinear', align_corners=False)
            panoptic, sem = self._materialised_head(fcn_output, pan_boxes, pan_logit, pan_cls, keep, num_keep, real_keep)
        elif fuse_up:
            panoptic, sem = ops.panoptic_fuse_up(fcn_score, 4, num_stuff, pan_boxes, pan_logit, pan_cls, keep, num_keep, real_keep, cmap)
        else:
            panoptic, sem = ops.panoptic_fuse(fcn_output, num_stuff, pan_boxes, pan_logit, pan_cls, keep, num_keep, real_keep, cmap,
                                              self.enable_void)
        k = int(num_keep.item())
        keep = keep[:k]
        self._tap(keep=keep, panoptic=panoptic, sem=sem)
        return {
            'cls_probs': det_scores, 'pred_boxes': det_boxes, 'mask_probs': mask_prob, 'fcn_outputs': sem,
            'cls_inds': det_cls, 'panoptic_cls_inds': pan_cls[keep], 'panoptic_cls_probs': pan_scores[keep],
            'panoptic_outputs': panoptic,
        }

    def _materialised_head(self, fcn_output, pan_boxes, pan_logit, pan_cls, keep, num_keep, real_keep):
        H, W = fcn_output.shape[2:]
        k = int(num_keep.item())
        energy = ops.mask_paste(pan_boxes[:, 1:], pan_logit, keep, num_keep, real_keep, k, (H, W))
        kk = keep[:k]
        _, seg_inst = self.seg_term(pan_cls[kk], fcn_output, pan_boxes[kk] * 4.0)
        num_stuff = self.num_seg_classes - (self.num_classes - 1)
        pan = ops.panoptic_argmax(fcn_output, num_stuff, seg_inst, energy, self.enable_void)
        return pan, torch.max(fcn_output, dim=1)[1]

    # ------------------------------------------------------------------ reference-shaped dataflow
    def _forward_modules(self, data):
        pyramid, rpn_cls_prob, rpn_bbox_pred = self._trunk(data)
        feats = list(pyramid[:4])
        im_info = data['im_info']
        rois, _ = self.pyramid_proposal(rpn_cls_prob, rpn_bbox_pred, im_info)
        fcn_output = self.fcn_head(*feats)
        rcnn_output = self.rcnn(feats, rois)
        cls_score, bbox_pred = rcnn_output['cls_score'], rcnn_output['bbox_pred']
        cls_prob = F.softmax(cls_score, dim=1)
        self._tap(rpn_cls_prob=rpn_cls_prob, rpn_bbox_pred=rpn_bbox_pred,
                  im_info=im_info, rois=rois, n_rois=rois.shape[0], cls_prob=cls_prob, bbox_pred=bbox_pred,
                  fcn_output=fcn_output['fcn_output'])

        cls_prob_all, mask_rois, cls_idx = self.mask_roi(rois, bbox_pred, cls_prob, im_info)
        mask_score = self.mask_branch(feats, mask_rois)
        mask_prob = torch.sigmoid(mask_score)
        results = {
            'cls_probs': cls_prob_all, 'pred_boxes': mask_rois, 'mask_probs': mask_prob,
            'fcn_outputs': torch.max(fcn_output['fcn_output'], dim=1)[1], 'cls_inds': cls_idx,
        }
        cls_prob, mask_rois, cls_idx = self.mask_roi_panoptic(rois, bbox_pred, cls_prob, im_info)
        mask_score = self.mask_branch(feats, mask_rois)
        ms = config.network.mask_size
        mask_score = mask_score.gather(1, cls_idx.view(-1, 1, 1, 1).expand(-1, -1, ms, ms))
        self._tap(det_boxes=results['pred_boxes'], det_scores=results['cls_probs'], det_cls=results['cls_inds'],
                  pan_boxes=mask_rois, pan_scores=cls_prob, pan_cls=cls_idx, pan_logit=mask_score)

        keep_inds, mask_logits = self.mask_removal(mask_rois[:, 1:], cls_prob, mask_score, cls_idx,
                                                   fcn_output['fcn_output'].shape[2:])
        mask_rois = mask_rois[keep_inds]
        cls_idx = cls_idx[keep_inds]
        cls_prob = cls_prob[keep_inds]
        seg_logits, seg_inst_logits = self.seg_term(cls_idx, fcn_output['fcn_output'], mask_rois * 4.0)
        results.update({'panoptic_cls_inds': cls_idx, 'panoptic_cls_probs': cls_prob})
        num_stuff = self.num_seg_classes - (self.num_classes - 1)
        results['panoptic_outputs'] = ops.panoptic_argmax(fcn_output['fcn_output'], num_stuff, seg_inst_logits,
                                                          mask_logits, self.enable_void)
        self._tap(keep=keep_inds, panoptic=results['panoptic_outputs'], sem=results['fcn_outputs'])
        return results


def resnet_101_upsnet(**kw):
    return resnet_upsnet([3, 4, 23, 3], **kw)


def resnet_50_upsnet(**kw):
    return resnet_upsnet([3, 4, 6, 3], **kw)
