"""One rank, torch.distributed (nccl) initialised first, then the library's own communicator through sharding.library_comm:
the path bench.py --gpus N takes on a multi-GPU node, as far as one GPU can show it."""
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
mi = importlib.import_module("lsp-dsp-units_amd")
from importlib import import_module
sharding = import_module("lsp-dsp-units_amd.sharding")
comm = sharding.library_comm(mi) or mi.Comm(mi.Comm.unique_id(), 1, 0)   # (library_comm: None for one rank)
print("library comm:", comm.info() if hasattr(comm, "info") else comm)
an = mi.AnalyzerBank(8, 8, 48000, 100.0)
an.configure(an.SAMPLE_RATE, 48000); an.configure(an.RANK, 8); an.configure(an.RATE, 187.5)
x = torch.randn((8, 256), device="cuda")
an.process(x, 256, stream=torch.cuda.current_stream()); an.process(x, 256, stream=torch.cuda.current_stream())
bins = torch.zeros((1, 144), device="cuda")
an.reduce_bins(bins[0], stream=torch.cuda.current_stream())
before = bins.clone()
an.allreduce_bins(bins, 1, comm, stream=torch.cuda.current_stream())
torch.cuda.synchronize()
print("allreduce over one rank leaves the sums:", bool(torch.equal(before, bins)), float(bins.abs().sum()) > 0)
comm.close(); an.close(); dist.destroy_process_group()
