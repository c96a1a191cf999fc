import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import fgoicp_amd as fg
def use_torch():
    import torch
    t = torch.arange(1024, device="cuda:0", dtype=torch.float32)
    assert float(t.sum()) == 523776.0
def use_rccl():
    ex = fg.RcclExchange(0, 1, fg.rccl_unique_id(), 0)
    assert ex.warmup()
    print("rccl:", fg._lib.load().fgoicp_rccl_library().decode(), flush=True)
    ex.close()
order = sys.argv[1]
for f in ((use_torch, use_rccl) if order == "torch_first" else (use_rccl, use_torch)):
    f()
print("done", flush=True)
