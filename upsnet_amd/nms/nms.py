"""NMS wrappers with the reference's names (upsnet/nms/nms.py:26-46, gpu_nms.pyx:23-38, cpu_nms.pyx:91-196).

All of them run the HIP kernels of csrc/nms.hip. ``gpu_nms_wrapper(thresh, device_id)(dets)`` takes a
float32 numpy array [N,5] (as the reference) or a CUDA tensor and returns a list of kept indices in
visiting order (score descending; equal scores: higher index first).
"""
import numpy as np
import torch

from .. import ops


def gpu_nms(dets, thresh, device_id=0):
    if isinstance(dets, torch.Tensor):
        return ops.gpu_nms(dets, thresh).tolist()
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    if dets.shape[0] == 0:
        return []
    t = torch.from_numpy(dets).to('cuda:%d' % device_id)
    return ops.gpu_nms(t, thresh).tolist()


def gpu_nms_wrapper(thresh, device_id):
    def _nms(dets):
        return gpu_nms(dets, thresh, device_id)
    return _nms


def _need_gpu(what):
    if not torch.cuda.is_available():
        raise RuntimeError("%s: upsnet_amd serves every NMS flavour with its HIP kernels and has no host path; no GPU is visible" % what)
    return torch.cuda.current_device()


def cpu_nms(dets, thresh):
    """cpu_nms (cpu_nms.pyx:29-80) on the device: suppression at ``overlap >= thresh`` (:77), thresh compared as a double."""
    dev = _need_gpu('cpu_nms')
    if isinstance(dets, torch.Tensor):
        return ops.cpu_nms(dets, thresh).tolist()
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    if dets.shape[0] == 0:
        return []
    return ops.cpu_nms(torch.from_numpy(dets).to('cuda:%d' % dev), thresh).tolist()


def cpu_nms_wrapper(thresh):
    def _nms(dets):
        return cpu_nms(dets, thresh)
    return _nms


def py_nms(dets, thresh):
    """py_nms (nms.py:48-85) keeps ``ovr <= thresh``, i.e. suppresses at ``>`` like the GPU kernel; its areas and overlaps are the
    same fp32 expressions -> served by the gpu_nms kernels."""
    return gpu_nms(dets, thresh, _need_gpu('py_nms'))


def py_nms_wrapper(thresh):
    def _nms(dets):
        return py_nms(dets, thresh)
    return _nms


def cpu_soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """Device soft-NMS with cpu_soft_nms' signature: numpy float32 [N,5] in -> (boxes', inds[:N'])."""
    is_np = not isinstance(boxes, torch.Tensor)
    t = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32)).cuda() if is_np else boxes
    b, inds = ops.soft_nms(t, sigma, Nt, threshold, method)
    if is_np:
        return b.cpu().numpy(), inds.cpu().numpy()
    return b, inds


def soft_nms_wrapper(thresh, method=1):
    def _nms(dets):
        return cpu_soft_nms(dets, Nt=thresh, method=method)
    return _nms
