"""`misc.data_parallel.DataParallelImbalance` for the fused engine (reference misc/data_parallel.py:70-112).

The reference wraps the model in this threaded single-process data-parallel module when it is NOT launched per GPU
(run_img2txt_dist.py:388-391, decode_img2txt.py, eval_vqa2.py).  The fused engine keeps its parameters, gradients and
activations in flat per-process device buffers and is driven one process per GPU (RCCL over xGMI:
vlp_amd.distributed.DistributedDataParallel), so the only configuration of the threaded module that has a meaning here
is the one the reference takes on a single visible GPU (:101-106): move the module to that device and pass the call
straight through.  More than one device raises and points at the replacement -- it never replicates silently."""
import torch
from torch import nn


class DataParallelImbalance(nn.Module):
    """Same constructor as misc/data_parallel.py:71 (module, device_ids=None, output_device=None, dim=0); exposes `.module`,
    `.device_ids`, `.output_device`, `.dim`."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super(DataParallelImbalance, self).__init__()
        self.dim = dim
        self.module = module
        if not torch.cuda.is_available():                    # :75-78
            self.device_ids = []
            self.output_device = None
            return
        if device_ids is None:                               # :80-81
            device_ids = list(range(torch.cuda.device_count()))
        device_ids = [d.index if isinstance(d, torch.device) else int(d) for d in device_ids]
        if len(device_ids) > 1:
            raise RuntimeError("vlp_amd: threaded single-process data parallelism over %d devices is not provided -- launch one "
                               "process per GPU (torchrun / --local_rank) and wrap the model in "
                               "vlp_amd.distributed.DistributedDataParallel (RCCL gradient buckets); or pass device_ids=[k]"
                               % len(device_ids))
        if output_device is None:                            # :82-83
            output_device = device_ids[0]
        self.device_ids = device_ids
        self.output_device = output_device.index if isinstance(output_device, torch.device) else int(output_device)
        if len(self.device_ids) == 1:                        # :95-96
            self.module.cuda(device_ids[0])

    def forward(self, *inputs, **kwargs):                    # :98-106: no scatter needed for one device
        return self.module(*inputs, **kwargs)
