"""GPU: pre-flight of the N > 1 path with what ONE GPU allows: the RCCL communicator (backend 'nccl'), the
gather-to-rank-0 code path on it, and the `bench.py --gpus N --dry-run` report (graph capture on both instances, replay == eager,
device memory per rank). Reference semantics: lib/utils/data_parallel.py:103-125 (gather to the first device),
upsnet/upsnet_end2end_test.py:224-247."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_gather_to_rank0_on_rccl_single_rank():
    """init_process_group('nccl') with world = 1 on the GPU, then the SAME collective sequence an 8-rank run executes (gather of the
    record counts, broadcast of the padded row count, gather of the payload) -- device tensors through RCCL."""
    from upsnet_amd.upsnet_end2end_test import gather_results
    assert not dist.is_initialized()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % (29700 + os.getpid() % 200), rank=0, world_size=1)
    try:
        assert dist.get_backend() == 'nccl'
        dev = torch.device('cuda', 0)
        local = [(i, torch.full((40, 64), i, dtype=torch.uint8, device=dev), 3 * i) for i in range(5)]
        out = gather_results(local, 1, dev, 48, 64, collective=True)
        assert sorted(out) == list(range(5))
        for i in range(5):
            lab, n = out[i]
            assert n == 3 * i and lab.is_cuda and int(lab[:40].min()) == i == int(lab[:40].max()) and int(lab[40:].min()) == 255
        empty = gather_results([], 1, dev, 48, 64, collective=True)
        assert empty == {}
        # the STREAMED form of the timed loop on RCCL (r10): rows known up front, int64 label maps written straight into the send buffers,
        # one asynchronous dist.gather per full chunk into row ranges of the preallocated result store, the ragged last chunk in finish()
        from upsnet_amd.upsnet_end2end_test import ResultGatherer
        g = ResultGatherer(1, dev, 48, 64, rows=11, chunk=4, collective=True)
        for s in range(11):
            g.add(100 + s, torch.full((48, 64) if s % 2 else (40, 60), s, dtype=torch.int64, device=dev), 7 * s)
            assert g.sent == (s + 1) // 4 * 4
        res = g.finish()
        assert sorted(res) == list(range(100, 111)) and g.collectives == 3 and g.store.is_cuda
        for s in range(11):
            lab, n = res[100 + s]
            h, w = (48, 64) if s % 2 else (40, 60)
            assert n == 7 * s and bool((lab[:h, :w] == s).all()) and (s % 2 or (int(lab[h:].min()) == 255 and int(lab[:, w:].min()) == 255))
        t = torch.tensor([2.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)        # the bench's max-over-ranks timing reduction
        dist.barrier()
        assert float(t) == 2.5
    finally:
        dist.destroy_process_group()


def _run_bench(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    return r, (json.loads(lines[-1]) if lines else None)


def test_dry_run_single_rank_reports_graphs_and_memory():
    r, rep = _run_bench(['--gpus', '1', '--dry-run'])
    assert r.returncode == 0 and rep is not None, r.stderr[-2000:]
    assert rep['dry_run'] and rep['all_ok'] and rep['ranks'] == 1 and rep['graph_instances_per_rank'] == 2
    me = rep['per_rank'][0]
    assert me['graphs_captured'] == 2 and me['graph_replay_equals_eager'] == 1
    # two graph instances of UPSNet-50 at 1024x2048 (activations + workspaces + weights): well inside one 288 GB device, and 8 ranks each
    # own a whole device, so the same footprint holds per rank on the 8-GPU node
    assert 1024 < me['torch_reserved_peak_mib'] < 64 * 1024, me


def test_dry_run_two_ranks_sharing_the_gpu():
    """The N > 1 launcher path end to end on a 1-GPU box: bench.py re-execs itself under torch.distributed.run with 2 ranks (free
    port on 127.0.0.1, per-rank CPU slices), both capture their graphs, gather to rank 0 (through gloo: RCCL refuses two ranks on
    one device), rank 0 prints the report."""
    r, rep = _run_bench(['--gpus', '2', '--dry-run'], {'UPSNET_SHARE_GPU': '1'}, timeout=1500)
    assert r.returncode == 0 and rep is not None, (r.stdout[-1000:], r.stderr[-3000:])
    assert rep['ranks'] == 2 and rep['all_ok'] and [p['rank'] for p in rep['per_rank']] == [0, 1]
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 2:
        assert all(p['host_cpus'] == ncpu // 2 for p in rep['per_rank']), rep['per_rank']
    g = rep['gather']                                   # what the final-gather path will move for the requested --steps (r10)
    assert g['chunk_images'] == 8 and g['bytes_sent_per_rank'] == g['steps'] * (1024 * 2048 + 16)
    assert g['rank0_in_flight_bytes_max'] == 2 * 8 * 1024 * 2048


def test_two_rank_timed_run_streams_its_label_maps_to_rank0():
    """The timed N = 2 path itself on a shared GPU (gloo staging): 11 images per rank -> one full chunk of 8 leaves while the loop is still
    running, the ragged rest in finish(); rank 0 reports n_gpus 2 and the gather statistics."""
    r, line = _run_bench(['--gpus', '2', '--steps', '11', '--warmup', '2', '--no-cpu-baseline', '--no-configs2', '--no-wide-offsets'],
                         {'UPSNET_SHARE_GPU': '1'}, timeout=1500)
    assert r.returncode == 0 and line is not None, (r.stdout[-1000:], r.stderr[-3000:])
    assert line['n_gpus'] == 2 and line['steps'] == 11 and line['value'] > 0
    g = line['config']['gather']
    assert g['chunk_images'] == 8 and g['collectives'] == 2 + 3 and g['bytes_sent'] == 2 * 8 * 1024 * 2048 + 16 * 11
    assert line['gather_s'] is not None and line['gather_s'] < line['timed_s']
