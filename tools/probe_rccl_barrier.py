"""One-off probe (not a test; pytest only collects tests/): cost of an RCCL barrier / gloo barrier / 1-element all-reduce on one rank."""


def main():
    import os, time, torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
    torch.cuda.set_device(0)
    dev=torch.device("cuda",0)
    dist.init_process_group(backend="nccl", device_id=dev)
    g=dist.new_group(backend="gloo")
    x=torch.zeros(1,device=dev)
    for name,fn in (("nccl barrier", lambda: dist.barrier()), ("gloo barrier", lambda: dist.barrier(group=g)), ("nccl all_reduce 1 elem + sync", lambda: (dist.all_reduce(x), torch.cuda.synchronize()))):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); print(name, (time.perf_counter()-t0)/50*1e6, "us")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
