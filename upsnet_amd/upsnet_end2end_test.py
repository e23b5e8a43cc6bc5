"""Inference driver: the MI355X counterpart of upsnet/upsnet_end2end_test.py:155-312.

The reference builds the model once, wraps it in a single-process threaded DataParallel that re-broadcasts
every parameter on every forward (lib/utils/data_parallel.py:103-125), feeds ONE image per GPU per
iteration, times `forward + torch.cuda.synchronize()` as `net_time` (skipping the first 10 iterations,
upsnet_end2end_test.py:244-252) and leaves the outputs on their GPUs.

Here: one PROCESS per GPU (torchrun), weights built/loaded once per rank, image i goes to rank
i mod world_size (independent units, no data-path collective), the same net_time window per image, and
ONE RCCL gather of the small per-image results (uint8 label maps + counters) to rank 0 after the loop.
Datasets, cv2 post-processing and PQ evaluation are out of scope;
inputs are the synthetic images of upsnet_amd.synthetic.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .config.config import CITYSCAPES_R50, COCO_R101_DCN, config, update_config_dict
from .synthetic import build_model, make_image, make_image_u8
from .utils.timer import Timer

WORKLOADS = {
    # name: (config preset, unpadded H, W, cls_gain)
    'upsnet50_cityscapes_1024x2048': (CITYSCAPES_R50, 1024, 2048, 'default'),
    'upsnet101dcn_coco_800x1333': (COCO_R101_DCN, 800, 1333, 'default'),
    # BASELINE.json configs[4]: alternating Cityscapes-shaped / COCO-shaped images through one UPSNet-101-DCN
    'upsnet101dcn_mixed_1024x2048_800x1333': (COCO_R101_DCN, (1024, 800), (2048, 1333), 'default'),
}


def rank_cpu_slice(local_rank, local_world, cpus=None):
    """The host CPUs of one rank: the visible CPUs (sched_getaffinity, sorted) cut into local_world contiguous slices. On a
    256-thread host with 8 ranks that is 32 hardware threads per rank, contiguous ids = one NUMA domain / CCD group per rank on the
    usual enumeration, so the 8 Python launch loops and their RCCL proxy threads do not migrate across sockets
    (lib/utils/data_parallel.py:103-116 runs all replicas on GIL-bound threads of ONE process instead)."""
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    per = max(1, len(cpus) // max(1, local_world))
    lo = (local_rank % max(1, len(cpus) // per)) * per
    return cpus[lo:lo + per]


def pin_rank(local_rank, local_world):
    """Pin this process to its CPU slice and size the host thread pools to it (no-op for a single rank, or with UPSNET_PIN=0)."""
    if local_world <= 1 or os.environ.get('UPSNET_PIN', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        return None
    mine = rank_cpu_slice(local_rank, local_world)
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    torch.set_num_threads(max(1, min(len(mine), int(os.environ.get('OMP_NUM_THREADS', len(mine))))))
    return mine


def init_distributed():
    """torchrun environment -> (rank, world, device). Backend nccl == RCCL on ROCm; gloo for CPU tests."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    pin_rank(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        ndev = torch.cuda.device_count()
        per_node = int(os.environ.get('LOCAL_WORLD_SIZE', world))
        if per_node > ndev:
            # one rank per GPU is the contract; several ranks on one GPU only on request (single-GPU functional runs of the N>1 path)
            if os.environ.get('UPSNET_SHARE_GPU', '0') != '1':
                raise RuntimeError('%d ranks on this node but only %d GPU(s) visible (set UPSNET_SHARE_GPU=1 to let ranks share a GPU)' % (per_node, ndev))
            local = local % ndev
            # RCCL refuses two ranks on one device ("Duplicate GPU detected"): EVERY rank of a shared-GPU run stages through gloo
            os.environ.setdefault('UPSNET_DIST_BACKEND', 'gloo')
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = os.environ.get('UPSNET_DIST_BACKEND', 'nccl' if use_cuda else 'gloo')
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, (torch.device('cuda', local) if use_cuda else torch.device('cpu'))


def shard_indices(num_images, rank, world):
    """image i -> rank i mod world (the reference's one-image-per-GPU round robin)."""
    return list(range(rank, num_images, world))


_REC_HDR = 16   # bytes in front of every label map: image id, instance count (2 x int64, little endian)


def pack_records(local, H, W, device):
    """[(image_id, label_map uint8 [h<=H, w<=W], n_inst)] -> one uint8 tensor [n, 16 + H*W] (header + row-major label map; the
    record outside a smaller label map is 255 = void, never class 0)."""
    rec = torch.full((len(local), _REC_HDR + H * W), 255, dtype=torch.uint8, device=device)
    rec[:, :_REC_HDR] = 0
    if local:
        hdr = torch.tensor([[i, n] for i, _, n in local], dtype=torch.int64).view(torch.uint8).view(len(local), _REC_HDR)
        rec[:, :_REC_HDR] = hdr.to(device)
    for j, (_, lab, _) in enumerate(local):
        assert lab.dtype == torch.uint8 and lab.shape[0] <= H and lab.shape[1] <= W
        rec[j, _REC_HDR:].view(H, W)[:lab.shape[0], :lab.shape[1]] = lab.to(device)
    return rec


def unpack_records(rec, H, W):
    out = {}
    if rec.shape[0]:
        hdr = rec[:, :_REC_HDR].cpu().reshape(-1).clone().view(torch.int64).view(-1, 2)
        for j in range(rec.shape[0]):
            out[int(hdr[j, 0])] = (rec[j, _REC_HDR:].view(H, W), int(hdr[j, 1]))
    return out


def gather_results(local, world, device, H=None, W=None, dst=0, collective=None):
    """The path's only collective (upsnet_end2end_test.py:224-247 leaves the outputs on their GPUs and the host collects them;
    data_parallel.py:103-116 gathers to the first device): every rank sends its per-image records -- local = list of
    (image_id, label_map uint8, n_inst) -- to rank `dst` ONLY. Returns a dict image_id -> (label_map [H,W], n_inst) on rank
    `dst` and **None on every other rank** (contract since r06: callers must not expect the result on all ranks). H, W = record shape (the workload's padded size; defaults to the first local map).
    One gather of the 8-byte record counts, then one gather of the payload (ranks with fewer records pad to the largest count:
    with image i on rank i mod world the counts differ by at most one)."""
    if H is None or W is None:
        assert local, 'gather_results: give H, W when a rank may hold no image'
        H, W = local[0][1].shape
    if collective is None:
        collective = world > 1      # (collective=True with world == 1: the N > 1 code path on a one-rank communicator -- pre-flight tests)
    if not collective:
        return unpack_records(pack_records(local, H, W, device), H, W)
    if dist.get_backend() == 'gloo':
        device = torch.device('cpu')
    rank = dist.get_rank()
    count = torch.tensor([len(local)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(count) for _ in range(world)] if rank == dst else None
    dist.gather(count, counts, dst=dst)
    # every rank needs the padded row count; it follows from the sharding rule without another collective only if the caller
    # sharded by i mod world -- do not assume it: broadcast the maximum (8 bytes)
    mx = torch.tensor([max(int(c) for c in counts)] if rank == dst else [0], dtype=torch.int64, device=device)
    dist.broadcast(mx, src=dst)
    mx = int(mx)
    rec = pack_records(local, H, W, device)
    if rec.shape[0] < mx:
        rec = torch.cat([rec, rec.new_zeros((mx - rec.shape[0], rec.shape[1]))], 0)
    recv = [torch.empty_like(rec) for _ in range(world)] if rank == dst else None
    dist.gather(rec, recv, dst=dst)
    if rank != dst:
        return None
    out = {}
    for r in range(world):
        out.update(unpack_records(recv[r][:int(counts[r])], H, W))
    return out


def preflight(workload='upsnet50_cityscapes_1024x2048', in_flight=2):
    """`bench.py --gpus N --dry-run`: everything a timed N-rank run does BEFORE its timed region, without timing anything -- device
    per rank, model build, graph capture of every input shape on every graph instance (2 per rank by default), one checked forward
    per instance, the communicator warm-up gather of the final record shape -- and a per-rank report of the device memory it took.
    Returns (on rank 0) a dict; raises on any rank if its graph replay disagrees with its eager forward."""
    rank, world, device = init_distributed()
    preset, H, W, gain = WORKLOADS[workload]
    sizes = list(zip(H, W)) if isinstance(H, (tuple, list)) else [(H, W)]
    Hm, Wm = max(h for h, _ in sizes), max(w for _, w in sizes)
    update_config_dict(preset)
    hp, wp = (int(np.ceil(v / 32.0) * 32) for v in (Hm, Wm))
    torch.cuda.reset_peak_memory_stats(device)
    model = build_model(cls_gain=gain, device=device)
    ok = True
    with torch.no_grad():
        for j, (h, w) in enumerate(sizes):
            data = make_image(h, w, seed=j, device=device)
            model.use_graph = False
            want = model(data)['panoptic_outputs'].clone()
            model.use_graph = True
            for _ in range(3 * model.graph_slots + model.graph_slots):
                got = model(data)['panoptic_outputs']
                ok = ok and bool(torch.equal(got, want))
    torch.cuda.synchronize(device)
    graphs = sum(1 for slots in model._graphs.values() for ent in slots['slots'] if 'graph' in ent)
    if world > 1:
        dummy = [(j * world + rank, torch.zeros((hp, wp), dtype=torch.uint8, device=device), 0) for j in range(2)]
        got = gather_results(dummy, world, device, hp, wp)
        ok = ok and (got is None if rank != 0 else len(got) == 2 * world)
        torch.cuda.synchronize(device)
    free, total = torch.cuda.mem_get_info(device)
    mine = torch.tensor([rank, int(ok), graphs, torch.cuda.max_memory_reserved(device) >> 20, (total - free) >> 20, total >> 20,
                         len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else 0], dtype=torch.int64,
                        device=device if (world == 1 or dist.get_backend() != 'gloo') else 'cpu')
    rows = [mine]
    if world > 1:
        rows = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, rows, dst=0)
        dist.barrier()
    if rank != 0:
        return None
    keys = ('rank', 'graph_replay_equals_eager', 'graphs_captured', 'torch_reserved_peak_mib', 'device_used_mib', 'device_total_mib', 'host_cpus')
    per_rank = [dict(zip(keys, (int(v) for v in r.tolist()))) for r in rows]
    return dict(dry_run=True, workload=workload, ranks=world, shapes=['%dx%d' % s for s in sizes], graph_instances_per_rank=model.graph_slots,
                all_ok=all(r['graph_replay_equals_eager'] == 1 and r['graphs_captured'] == len(sizes) * model.graph_slots for r in per_rank),
                per_rank=per_rank)


def upsnet_test(workload='upsnet50_cityscapes_1024x2048', steps=20, warmup=10, seed=0, pipeline='fused', gather=True,
                on_step=None, on_warmup_done=None, input_mode='f32', post=False, in_flight=2, before_step=None):
    """Run `steps` timed images per rank (after `warmup` untimed ones). Returns a dict with the whole-job
    wall time (max over ranks, barrier + sync bracketed), per-image net_time samples and the gathered results.
    input_mode 'f32': the fp32 blob is resident in HBM (the benchmark workload); 'u8': the uint8 image is resident and the
    input kernel (dataset/blob.py) runs inside every step. post: get_unified_pan_result (dataset/base_dataset.py) runs inside
    every step too (the reference does it after the loop, on the host). in_flight: images launched per rank before the oldest
    one is read back (1 = strictly one after the other, like the reference's loop; 2 = the next launch overlaps the read-back)."""
    rank, world, device = init_distributed()
    preset, H, W, gain = WORKLOADS[workload]
    sizes = list(zip(H, W)) if isinstance(H, (tuple, list)) else [(H, W)]
    H, W = max(h for h, _ in sizes), max(w for _, w in sizes)   # (label maps are padded to the largest size for the gather)
    update_config_dict(preset)
    hp, wp = (int(np.ceil(v / 32.0) * 32) for v in (H, W))   # record shape of the final gather (inputs are padded to 32)
    model = build_model(cls_gain=gain, device=device, pipeline=pipeline)
    # each rank owns its images: image id = step * world + rank, seeded by id
    my_ids = [s * world + rank for s in range(steps)]
    if input_mode == 'u8':
        from .dataset.blob import get_image_blob
        pool = [make_image_u8(*sizes[j % len(sizes)], seed=seed + j, device=device) for j in range(4)]  # 4 distinct uint8 images resident in HBM

        def get(i):
            return get_image_blob(pool[i % 4], config.test.scales[0], config.test.max_size)
    else:
        pool = [make_image(*sizes[j % len(sizes)], seed=seed + j, device=device) for j in range(4)]  # 4 distinct fp32 blobs resident in HBM

        def get(i):
            return pool[i % 4]
    if post:
        from .dataset.base_dataset import BaseDataset
        post_fn = BaseDataset().get_unified_pan_result

    net_timer = Timer()
    outs = []
    with torch.no_grad():
        # one-off preparation, like the weight packing inside build_model: the HIP graph of each input shape is captured on
        # the third forward of that shape (two eager forwards first: packed weights, kernel attributes, library handles)
        if getattr(model, 'use_graph', False):
            for j in range(len(sizes)):
                for _ in range(3 * getattr(model, 'graph_slots', 1)):
                    model(get(j))
        for w in range(warmup):
            model(get(w))
        torch.cuda.synchronize(device)
        if world > 1 and gather:
            # warm the communicator with a gather of the final shape (RCCL sets up its channels / buffers on first use of a
            # collective at a given size; that one-off cost belongs to start-up, not to the timed loop)
            dummy = [(j * world + rank, torch.zeros((hp, wp), dtype=torch.uint8, device=device), 0) for j in range(steps)]
            gather_results(dummy, world, device, hp, wp)
            del dummy
            torch.cuda.synchronize(device)
        if on_warmup_done is not None:
            on_warmup_done(model)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        # `in_flight` images per rank: image i+1 is launched (one graph replay on that graph instance's own stream) before image
        # i's outputs are read back, so (a) the host work between two images -- read-back, Python, the next launch -- is off the
        # device's critical path and (b) the kernels of the two images overlap on the device (tail rounds, small layers and the
        # latency-bound detection chain of one are filled by the other). Every image is fully processed inside the timed region.
        # net_time = (completion of image i+d) - (completion of image i), divided by d = in_flight: the steady-state time per
        # image; lat_time = launch -> outputs on the host side.
        done_at, lat = [], []

        def finish(s, i, handle, t_launch):
            out = handle.result() if hasattr(handle, 'result') else handle
            if post:
                out['pan_2ch'] = post_fn([out['fcn_outputs']], [out['panoptic_outputs']], [out['panoptic_cls_inds']])[0]
            lab = out['panoptic_outputs'][0].to(torch.uint8)
            if len(sizes) > 1:  # mixed stream: common shape for the gather (255 = void outside the image)
                lab = torch.nn.functional.pad(lab, (0, wp - lab.shape[1], 0, hp - lab.shape[0]), value=255)
            outs.append((i, lab, int(out['panoptic_cls_inds'].numel())))
            if on_step is not None:
                on_step(s, out, model)
            if in_flight <= 1:
                torch.cuda.synchronize(device)
            now = time.perf_counter()
            done_at.append(now)
            lat.append(now - t_launch)
            return out

        depth = max(1, min(int(in_flight), getattr(model, 'graph_slots', 1))) if hasattr(model, 'forward_async') else 1
        launch = model.forward_async if depth > 1 else model
        pending = []
        done_at.append(time.perf_counter())
        for s, i in enumerate(my_ids):
            if before_step is not None:
                before_step(s, model)
            pending.append((s, i, launch(get(i)), time.perf_counter()))
            if len(pending) >= depth:
                out = finish(*pending.pop(0))
        while pending:
            out = finish(*pending.pop(0))
        net_timer.samples = [(done_at[k + depth] - done_at[k]) / depth for k in range(len(done_at) - depth)]
        results = gather_results(outs, world, device, hp, wp) if gather else None   # rank 0 only; None elsewhere
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return dict(rank=rank, world=world, elapsed=float(t.item()), net_times=list(net_timer.samples), latencies=list(lat), results=results,
                last_out=out, model=model, image=get(my_ids[-1]), H=H, W=W)


def main(argv=None):
    """Entry point with the reference's command line (upsnet_end2end_test.py:155-290: --cfg yaml [--weight_path ...]) over the
    synthetic workloads: builds the model from config.symbol / the workload preset, runs the process-per-GPU loop and logs the
    reference's line `Batch i/N, data_time, net_time, post_time`. Launch under torch.distributed.run for N > 1 ranks."""
    import logging
    from .config.parse_args import parse_args
    args = parse_args('UPSNet inference on MI355X (synthetic inputs)', argv)
    logging.basicConfig(level=logging.INFO, format='%(asctime)-15s | %(message)s')
    res = upsnet_test(args.workload, steps=args.steps, warmup=args.warmup, in_flight=args.in_flight)
    if res['rank'] == 0:
        nt = sorted(res['net_times'])
        n_img = args.steps * res['world']
        logging.info('Batch %d/%d, data_time:%.3f, net_time:%.3f, post_time:%.3f' % (n_img, n_img, 0.0, nt[len(nt) // 2] if nt else 0.0, 0.0))
        logging.info('%d ranks, %.2f images/sec (whole job), %d label maps gathered on rank 0' %
                     (res['world'], n_img / res['elapsed'], len(res['results'] or {})))
    if dist.is_initialized():
        dist.destroy_process_group()
    return res


if __name__ == '__main__':
    main()
