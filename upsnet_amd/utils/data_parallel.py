"""DataParallel with the reference's constructor and call convention (lib/utils/data_parallel.py:78-125) for the inference loop of
upsnet_end2end_test.py:203-247:

    test_model = DataParallel(test_model, device_ids=gpus, gather_output=False).to(gpus[0])
    output = test_model(*batch)          # batch = [(data, None), ...], one tuple of forward() arguments per GPU

What it computes is the reference's: one device -> `module(*inputs[0])` (its result, not a list); several devices -> one result per
input, in order (`gather_output=False`), or their dim-0 gather on `output_device`.

How it is built differs, because the reference's per-call `replicate()` re-broadcasts every parameter each step and runs the replicas
on GIL-bound Python threads: here the replicas are built ONCE (at construction / on first use: a deep copy of the
prepared module per extra device, packed weights and HIP graphs owned per replica) and a call launches every replica's forward
asynchronously from the calling thread (`forward_async`: one input copy + one hipGraphLaunch per device), then collects the results in
order -- no threads, no per-step broadcast. The throughput path for a whole node remains one PROCESS per GPU
(upsnet_amd/upsnet_end2end_test.py, bench.py --gpus N); this class is the drop-in for callers written against the reference's loop.
"""
import copy

import torch
from torch.nn.modules import Module


class DataParallel(Module):

    def __init__(self, module, device_ids=None, output_device=None, dim=0, gather_output=True):
        super(DataParallel, self).__init__()
        self.module = module
        self.dim = dim
        self.gather_output = gather_output
        if not torch.cuda.is_available():
            self.device_ids = []
            return
        if device_ids is None:
            device_ids = list(range(torch.cuda.device_count()))
        self.device_ids = [int(d) for d in device_ids]
        self.output_device = self.device_ids[0] if output_device is None else output_device
        self._replicas = None
        if len(self.device_ids) == 1:
            self.module.cuda(self.device_ids[0])

    def _ensure_replicas(self, n):
        """module on device_ids[0] + one prepared deep copy per further device, built once."""
        if self._replicas is None:
            self._replicas = [self.module.cuda(self.device_ids[0])]
        while len(self._replicas) < n:
            dev = self.device_ids[len(self._replicas)]
            rep = copy.deepcopy(self.module).cuda(dev)   # (captured graphs and packed weights are per module instance, never copied)
            if hasattr(rep, 'invalidate_graphs'):
                rep.invalidate_graphs()
            self._replicas.append(rep)
        return self._replicas[:n]

    def forward(self, *inputs, **kwargs):
        if not self.device_ids:
            return self.module(*inputs, **kwargs)
        assert kwargs == {}, 'not implemented'
        if len(self.device_ids) == 1:
            return self.module(*inputs[0])
        if len(inputs) > len(self.device_ids):
            raise ValueError('DataParallel: %d inputs for %d devices' % (len(inputs), len(self.device_ids)))
        replicas = self._ensure_replicas(len(inputs))
        pending = []
        for rep, dev, args in zip(replicas, self.device_ids, inputs):
            with torch.cuda.device(dev):
                launch = getattr(rep, 'forward_async', None)
                # forward(data, label=None): the asynchronous launch takes the data dict only
                if launch is not None and len(args) >= 1 and all(a is None for a in args[1:]):
                    pending.append(launch(args[0]))
                else:
                    pending.append(rep(*args))
        outputs = []
        for dev, h in zip(self.device_ids, pending):
            with torch.cuda.device(dev):
                outputs.append(h.result() if hasattr(h, 'result') else h)
        if self.gather_output:
            return self.gather(outputs, self.output_device)
        return outputs

    def gather(self, outputs, output_device):
        from torch.nn.parallel.scatter_gather import gather
        return gather(outputs, output_device, dim=self.dim)
