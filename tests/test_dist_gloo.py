"""N>1 data-parallel plumbing on CPU with the gloo backend (world_size 2): bucketed gradient reducer, broadcast of the initial
weights, packed SyncBN statistics."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from iseg_amd import dist, nn
    from iseg_amd.backbones import convnext as cx
    from iseg_amd.param_store import ParamStore

    nn.set_device("cpu")
    dist.init(backend="gloo")
    assert dist.world_size() == world and dist.rank() == rank
    nn.set_seed(rank)                     # different init per rank -> broadcast must equalise
    with nn.dry_run_scope():
        net = cx.ConvNeXt(depths=[1, 1, 1, 1], filters_list=[8, 16, 32, 64])
        net(torch.empty(1, 32, 32, 3))
    st = ParamStore(list(net.parameters()))
    st.broadcast_from_rank0()
    ref = st.flat_w.clone()
    dist.all_reduce_sum(ref)
    ok_bcast = torch.allclose(ref, st.flat_w * world)
    # bucketed reducer: tiny buckets so that several are in flight
    red = dist.GradReducer(st, bucket_bytes=8 << 10)
    assert len(red.buckets) > 3
    params = list(reversed(st.params))              # backward order
    shared = params[0]                              # a parameter the "model" uses twice per step (reports twice)
    b_shared = red.bucket_of[id(shared)]
    want = float(sum(range(1, world + 1)))
    ok_red = True
    for step in range(3):
        for p in st.params:
            p.grad.fill_(float(rank + 1))
        red.ready(shared)
        # step 0 only counts uses; later the bucket of `shared` must wait for the second report even when all others are in
        red.ready(*params[1: len(params) // 2])
        ok_red = ok_red and not red.launched[b_shared]
        if step > 0:
            ok_red = ok_red and any(red.launched)   # buckets completed during "backward" are already on the wire
        red.ready(shared)
        if step > 0:
            ok_red = ok_red and red.launched[b_shared]
        red.finish()                                # sweeps the buckets whose parameters never reported
        ok_red = ok_red and all(bool((p.grad == want).all()) for p in st.params)
    # a report that arrives after its bucket went out must raise, not corrupt the sum
    for p in st.params:
        p.grad.fill_(float(rank + 1))
    red.ready(shared)
    red.ready(*params[1: len(params) // 2])
    red.ready(shared)
    try:
        red.ready(shared)
        ok_red = False
    except RuntimeError:
        pass
    red.finish()
    # packed SyncBN message: [sum, sumsq, count]
    x = torch.arange(8, dtype=torch.float32).reshape(4, 2) + 10 * rank
    packed = torch.cat([x.sum(0), (x * x).sum(0), torch.tensor([4.0])])
    dist.all_reduce_sum(packed)
    allx = torch.cat([torch.arange(8, dtype=torch.float32).reshape(4, 2) + 10 * r for r in range(world)])
    n = packed[4]
    mean = packed[:2] / n
    var = packed[2:4] / n - mean * mean
    ok_bn = torch.allclose(mean, allx.mean(0)) and torch.allclose(var, allx.var(0, unbiased=False), rtol=1e-4)
    q.put((rank, ok_bcast, ok_red, ok_bn))
    dist.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_reducer_broadcast_syncbn():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, ok_bcast, ok_red, ok_bn in res:
        assert ok_bcast and ok_red and ok_bn, (rank, ok_bcast, ok_red, ok_bn)
