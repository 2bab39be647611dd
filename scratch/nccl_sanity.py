import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=torch.frombuffer(bytearray(b"abcdefgh"*29), dtype=torch.uint8).to("cuda")
outs=[torch.empty_like(t)]
dist.all_gather(outs,t); dist.barrier(); torch.cuda.synchronize()
print("nccl ok", bytes(outs[0].cpu().numpy().tobytes())[:8])
dist.destroy_process_group()
