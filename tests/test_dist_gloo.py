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


def _worker(rank, world, port, q, exchange=""):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if exchange:      # "emulate": the stream-ordered exchange's scheduling (dist._launch_stream_ordered, the join in finish()) over a blocking gloo call
        os.environ["ISEG_DIST_NATIVE"] = exchange
    else:
        os.environ.pop("ISEG_DIST_NATIVE", None)
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from iseg_amd import dist, nn
    from iseg_amd.backbones import convnext as cx
    from iseg_amd.param_store import ParamStore

    nn.set_device("cpu")
    dist.init(backend="gloo")
    assert dist.world_size() == world and dist.rank() == rank
    assert dist.native_mode() == exchange
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


import pytest  # noqa: E402


@pytest.mark.parametrize("exchange", ["", "emulate"])
def test_two_rank_gloo_reducer_broadcast_syncbn(exchange):
    """exchange = "": c10d work objects (the default data-parallel path); "emulate": the stream-ordered exchange of ISEG_DIST_NATIVE=1 -- the same
    bucket scheduling, launch points and join the RCCL-through-the-C-ABI path takes on a GPU -- with gloo as the primitive"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    for rank, ok_bcast, ok_red, ok_bn in res:
        assert ok_bcast and ok_red and ok_bn, (rank, ok_bcast, ok_red, ok_bn)


def _eval_worker(rank, world, port, q):
    """evaluate() under two ranks with RAGGED shards (3 batches -> rank 0 gets two, rank 1 gets one) and the progress line switched on:
    every collective must be entered by every rank the same number of times (a rank-0-only metric read used to deadlock here).  The device
    kernels are replaced by torch-CPU stand-ins INSIDE THIS TEST PROCESS only: what is under test is the collective protocol."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from iseg_amd import dist, nn
    from iseg_amd.core_model import SegFoundation
    from iseg_amd.data import synthetic_dataset
    from iseg_amd.distribution.distribution_utils import Strategy
    from iseg_amd.evaluations import evaluation as E
    from iseg_amd.metrics.mean_iou import MeanIOU

    nn.set_device("cpu")
    dist.init(backend="gloo")

    class Stub(SegFoundation):
        def __init__(self):
            torch.nn.Module.__init__(self)

        def inference_with_multi_scales(self, images, training=False, scale_rates=(1.0,), flip=False):
            # deterministic "logits": class = a function of the pixel values
            c = (images.sum(-1, keepdim=True) * 3.0).floor().long() % 21
            return torch.nn.functional.one_hot(c.squeeze(-1), 21).float() * 4.0

    def cpu_loss(num_class=21, ignore_label=255, batch_size=2, reduction=False, **kw):
        def fn(y_true, y_pred):
            z = y_pred.reshape(-1, num_class)
            y = y_true.reshape(-1).long()
            keep = y != ignore_label
            lp = torch.log_softmax(z, -1)
            return torch.where(keep, -lp[torch.arange(len(y)), torch.where(keep, y, torch.zeros_like(y))], torch.zeros(len(y)))
        return fn

    def cpu_confusion(self, logits2d, labels1d, ignore_label):
        pred = logits2d.argmax(-1)
        keep = labels1d != ignore_label
        idx = labels1d[keep].long() * self.num_classes + pred[keep]
        self.total_cm += torch.bincount(idx, minlength=self.num_classes ** 2)

    E.catecrossentropy_ignore_label_loss = cpu_loss
    MeanIOU.update_from_logits = cpu_confusion
    data = synthetic_dataset(5, 16, 16, seed=3)      # batch 2 -> 3 batches: ranks hold 2 and 1
    miou = float(E.evaluate(Strategy(one_device=(world == 1)), Stub(), data, batch_size=2, num_class=21, ignore_label=255, scale_rates=[1.0],
                            flip=False, val_image_count=5, verbose=1))
    q.put((rank, miou, E.evaluate.last_mean_loss))
    if dist.is_initialized():
        torch.distributed.destroy_process_group()


def test_two_rank_evaluate_enters_every_collective_on_every_rank():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, "a rank hung or failed inside evaluate()"
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]      # both ranks report the merged figures
    # single-process reference over the same five images
    port1 = _free_port()
    q1 = ctx.Queue()
    p = ctx.Process(target=_eval_worker, args=(0, 1, port1, q1))
    p.start()
    p.join(timeout=240)
    assert p.exitcode == 0
    _, miou1, loss1 = q1.get(timeout=5)
    assert abs(res[0][1] - miou1) < 1e-9 and abs(res[0][2] - loss1) < 1e-6


def _comm_order_worker(rank, world, port, q):
    """rank 1 asks for its gradient-bucket communicator FIRST, rank 0 for the SyncBN one: both must end up with the same id on "comm" and the same id
    on "comm_side" (creation = one c10d broadcast per communicator; a fake C ABI records the ids, no RCCL on this box)"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ctypes as C
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from iseg_amd import _hip, dist

    class FakeLib:
        def __init__(self):
            self.n = 0
            self.inits = []

        def iseg_comm_unique_id(self, buf):
            self.n += 1
            buf.raw = bytes([self.n]) * 128      # rank 0 draws id 1 for its first communicator, id 2 for the second
            return 0

        def iseg_comm_init(self, comm_ref, world_size, rank_, uid):
            self.inits.append(uid[0])
            comm_ref._obj.value = 1000 + uid[0]      # the "handle" carries the id it was created from
            return 0

    fake = FakeLib()
    _hip.lib = lambda: fake
    _hip.check = lambda code, what: None
    dist.init(backend="gloo")
    first, second = ("comm", "comm_side") if rank == 0 else ("comm_side", "comm")
    a = dist._native_comm(first)
    b = dist._native_comm(second)
    ids = {first: a.value, second: b.value}
    q.put((rank, ids["comm"], ids["comm_side"], fake.inits))
    import torch.distributed as td

    td.barrier()
    td.destroy_process_group()


def test_native_communicators_come_up_in_one_order():
    """round-5 verdict item 6: the C-ABI communicators ("comm": SyncBN messages, "comm_side": gradient buckets) are created in the same order on
    every rank even when a rank reaches its first bucket before its first SyncBN message"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_comm_order_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, c0, s0, i0), (_, c1, s1, i1) = res
    assert (c0, s0) == (1001, 1002) and (c1, s1) == (1001, 1002), res      # same id behind the same name on both ranks
    assert i0 == [1, 2] and i1 == [1, 2], res                               # and created in the same order
