import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.full((5,), float(rank + 1), dtype=torch.float64, device="cuda:0")
    dist.all_reduce(t)
    mn = torch.tensor([float(rank + 3)], dtype=torch.float64, device="cuda:0")
    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    print("rank", rank, t.tolist(), mn.tolist(), flush=True)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(w, args=(2,), nprocs=2)
