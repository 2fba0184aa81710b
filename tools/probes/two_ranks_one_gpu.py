"""r05 probe: can RCCL build a TWO-rank communicator with both ranks on the one GPU of a gpurun box (so that bogp_exchange_* could meet a second rank without an
8-GPU node)?  No: ncclCommInitRank returns "invalid usage" (duplicate GPU) on both ranks -- which the library reports as BOGP_ERR_HIP with the message, and bench.py answers
with its torch.distributed fall-back.  The exchange has run on hardware with world = 1 only."""
import os, sys, time, multiprocessing as mp
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
def worker(rank, q):
    import numpy as np
    from bogp import _lib
    eng = _lib.Engine(0)
    if rank == 0:
        uid = _lib.comm_unique_id(); open("/tmp/uid.bin", "wb").write(uid)
    else:
        while not os.path.exists("/tmp/uid.bin") or os.path.getsize("/tmp/uid.bin") == 0: time.sleep(0.05)
        time.sleep(0.2); uid = open("/tmp/uid.bin", "rb").read()
    try:
        eng.comm_init(uid, rank, 2)
        q.put((rank, "comm_init ok", eng.comm_world))
    except Exception as e:
        q.put((rank, "comm_init failed: %r" % (e,), None))
if __name__ == "__main__":
    if os.path.exists("/tmp/uid.bin"): os.remove("/tmp/uid.bin")
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, q)) for r in range(2)]
    for p in ps: p.start()
    for _ in ps:
        try: print(q.get(timeout=60))
        except Exception as e: print("timeout", e)
    for p in ps:
        p.join(5)
        if p.is_alive(): p.kill()
