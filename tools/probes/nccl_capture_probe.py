"""GPU probe: does a hipGraph stream capture on stream S disturb RCCL work whose end event was recorded on S?
(ProcessGroupNCCL's watchdog thread polls hipEventQuery on that event while the host is capturing.)
    python tools/probes/nccl_capture_probe.py same|comm
"""
import os, sys, time
import torch
mode = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
S, C = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.ones(5_000_000, device=dev)
x = torch.ones(1024, device=dev)
for it in range(3):
    with torch.cuda.stream(S):
        x.mul_(1.0001)
        if mode == "same":
            torch.distributed.all_reduce(g)
        else:
            C.wait_stream(S)
            with torch.cuda.stream(C):
                torch.distributed.all_reduce(g)
            S.wait_stream(C)
        x.add_(1.0)
        graph = torch.cuda.CUDAGraph()
        # capture on S for a while (the watchdog polls every ~100 ms)
        with torch.cuda.graph(graph, stream=S, capture_error_mode="relaxed"):
            for _ in range(50):
                x.mul_(1.0)
            time.sleep(0.5)
        graph.replay()
    torch.cuda.synchronize()
    print(mode, "iteration", it, "ok", flush=True)
time.sleep(1.0)
torch.distributed.destroy_process_group()
print(mode, "done", flush=True)
