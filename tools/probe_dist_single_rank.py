import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
ge.build()
from metagraph_amd import aligner, capi, synth
dev = torch.device("cuda", 0)
k, L, n = 31, 150, 20000
genome = synth.random_genome(2_000_000, 1, dev)
boss = synth.build_boss([genome[None, :]], k)
W, last = boss["W"].contiguous(), boss["last"].contiguous()
G = aligner.Graph(k, (W.data_ptr(), boss["n_edges"] + 1), (last.data_ptr(), boss["n_edges"] + 1), boss["F"], device=0, on_device=True)
reads = synth.sample_reads(genome, n, L, 2).contiguous()
offsets = (torch.arange(n + 1, device=dev, dtype=torch.int64) * L).contiguous()
A = aligner.Aligner(G, capi.config_cli(k))
A.align_device(reads.data_ptr(), offsets.data_ptr(), n)
lib = capi.lib()
hp, hb, nq, sp, sw = C.c_void_p(), C.c_uint64(), C.c_uint64(), C.c_void_p(), C.c_uint64()
rc = lib.mgx_device_results(A.h, C.byref(hp), C.byref(hb), C.byref(nq), C.byref(sp), C.byref(sw))
assert rc == 0
n_bytes = hb.value * nq.value
class _Ptr:
    __cuda_array_interface__ = {"shape": (n_bytes,), "typestr": "|u1", "data": (hp.value, False), "version": 2}
hdr = torch.as_tensor(_Ptr(), device=dev)
print("record bytes", hb.value, "reads", nq.value, "tensor", hdr.shape, hdr.dtype, hdr.device, "checksum", int(hdr.to(torch.int64).sum()))
# single-rank process group over RCCL: init + gather to self + barrier
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
import torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
gl = [torch.empty_like(hdr)]
dist.gather(hdr, gl, dst=0)
dist.barrier()
torch.cuda.synchronize()
print("gather ok", bool((gl[0] == hdr).all()))
t = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); print("allreduce", float(t))
dist.destroy_process_group()
