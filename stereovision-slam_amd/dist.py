"""One process per GPU: rank discovery, stream sharding and timing reduction.

The hot path shards by stream (frames of one stream are strictly sequential,
streams are independent), so there is NO data-path collective: ranks only meet at
the timing barrier and at one max-reduction of the elapsed time.  `backend` is
"nccl" (= RCCL over xGMI on ROCm) on GPUs and "gloo" in the CPU tests."""
import os


class Rank:
    def __init__(self, rank, local_rank, world, dist=None, device=None):
        self.rank, self.local_rank, self.world, self.dist, self.device = rank, local_rank, world, dist, device

    @property
    def is_root(self):
        return self.rank == 0

    def stream_seeds(self, streams_per_rank, base=0x5EED0000):
        """global stream ids owned by this rank (seed-addressed synthetic streams)"""
        first = self.rank * streams_per_rank
        return [base + first + s for s in range(streams_per_rank)]

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def allreduce(self, arr, op="sum"):
        """element-wise sum / max of a float64 numpy vector over the ranks (RCCL on GPUs, gloo in the CPU tests)"""
        import numpy as np
        a = np.ascontiguousarray(arr, np.float64)
        if self.dist is None:
            return a.copy()
        import torch
        t = torch.from_numpy(a.copy())
        if self.device:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX)
        return t.cpu().numpy()

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def init(backend="nccl"):
    """reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract)"""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # A lone process needs no process group.  Under torch.distributed.run (what the driver launches N > 1 with, and what
    # tools/scale.sh launches at any N) the group is formed even for one rank, so that the RCCL path — communicator,
    # barrier, all-reduce on the device — runs on every box, not only where several GPUs are visible.
    if world <= 1 and "TORCHELASTIC_RUN_ID" not in os.environ:
        return Rank(0, local_rank, 1)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = None
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        device = "cuda"
    dist.init_process_group(backend, rank=rank, world_size=world)
    return Rank(rank, local_rank, world, dist, device)


def _parse_cpulist(txt):
    out = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def device_numa_node(device_index):
    """NUMA node the GPU hangs off (sysfs, via the device's PCI address); -1 when it cannot be told (no GPU, no sysfs entry)"""
    try:
        import torch
        if not torch.cuda.is_available():
            return -1
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        return int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return -1


def device_numa_cpus(device_index, physical_only=True):
    """CPUs of the NUMA node the GPU hangs off (sysfs, via the device's PCI address); with
    physical_only the first hardware thread of each core.  Empty set when it cannot be told."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return set()
        cpus = _parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read())
        if physical_only:
            first = set()
            for c in cpus:
                sib = _parse_cpulist(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read())
                first.add(min(sib))
            cpus &= first
        return cpus
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return set()


def pin_to_device_numa(device_index, min_cpus=1):
    """Restrict this process (and the threads it creates from now on) to the GPU's NUMA node.
    The host side of the path is a random walk over per-stream maps: keeping its threads and their
    memory on one socket cut the CPU time per frame by ~15 % on a 2-socket EPYC host
    (tools/numa.sh).  No-op unless at least min_cpus allowed CPUs remain.  Returns the CPU set used."""
    if not hasattr(os, "sched_setaffinity"):
        return set()
    allowed = os.sched_getaffinity(0)
    cpus = device_numa_cpus(device_index) & allowed
    if len(cpus) < max(1, min_cpus):
        cpus = device_numa_cpus(device_index, physical_only=False) & allowed
    if len(cpus) < max(1, min_cpus):
        return set()
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return set()
    return cpus
