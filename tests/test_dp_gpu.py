"""Data-parallel semantics on the GPU path: two ranks (both on cuda:0, gloo transport because one box has one GPU) with half
the batch each must reproduce the single-process step on the full batch -- SyncBN statistics (packed all-reduce, forward and
backward), bucketed gradient all-reduce, 1/world gradient scaling, identical post-step weights on both ranks."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _step(rank, world, port, q, size, batch, rccl_single=False, native=False, graphed=False, steps=2):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if world > 1:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                          ISEG_DIST_BACKEND="gloo")
    else:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ISEG_DIST_BACKEND"):
            os.environ.pop(k, None)
        if rccl_single:     # one rank, every collective issued for real on the RCCL backend
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                              ISEG_DIST_SINGLE_RANK_COLLECTIVES="1", ISEG_DIST_BACKEND="nccl")
    if native:      # the stream-ordered exchange: every collective through the C ABI's own RCCL communicator (iseg_amd/dist.py)
        os.environ["ISEG_DIST_NATIVE"] = "1"
    else:
        os.environ.pop("ISEG_DIST_NATIVE", None)
    from iseg_amd import dist, nn
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.data import synthetic_batch
    from iseg_amd.distribution.distribution_utils import Strategy
    from iseg_amd.heads import convnext_tiny_aspp
    from iseg_amd.trainer import TrainableModel
    from tests.util_models import randomize_parameters

    nn.set_compute_dtype(torch.float32)
    strat = Strategy(one_device=(world == 1 and not rccl_single))
    if rccl_single:
        assert dist.active() and torch.distributed.get_backend() == "nccl" and dist.world_size() == 1
    model = convnext_tiny_aspp(num_class=21, build_input_size=size, drop_path_rate=0.0, dropout_rate=0.0, layer_scale_init_value=1.0)
    randomize_parameters(model, 0)
    opt = get_optimizer(strat, initial_lr=1e-3, epoch_steps=10, train_epoch=1, optimizer="sgd", sgd_momentum_rate=0.9)
    tm = TrainableModel(model, optimizer=opt, loss=model.custom_losses(21, 255, batch), loss_weights=model.custom_losses_weights(),
                        metrics=model.custom_metrics(21, 255))
    randomize_parameters(model, 0)      # after the ParamStore took over the storage
    x, y = synthetic_batch(batch, size[0], size[1], seed=3)
    per = batch // world
    xs, ys = x[rank * per:(rank + 1) * per].cuda(), y[rank * per:(rank + 1) * per].cuda()
    sizes = []
    real_allreduce = dist.all_reduce_sum
    dist.all_reduce_sum = lambda t, async_op=False: (sizes.append(int(t.numel())), real_allreduce(t, async_op=async_op))[1]
    step_fn = tm.train_step
    if graphed:
        from iseg_amd.graphs import GraphedTrainStep

        step_fn = GraphedTrainStep(tm, warmup=1)
        assert step_fn._eligible(xs), "the data-parallel step must be capturable with the stream-ordered exchange"
    losses = [float(step_fn(xs, ys)[0]) for _ in range(steps)]
    dist.all_reduce_sum = real_allreduce
    if graphed:
        assert any(e.get("graph") is not None for e in step_fn.entries.values()), "the step was never captured"
        sizes = None      # (replays issue no host-side calls: the message census below belongs to the eager runs)
    if native:
        assert dist.native_mode() == "rccl" and dist._NATIVE["comm"] is not None, "no collective went through the C-ABI communicator"
    torch.cuda.synchronize()
    if (world > 1 or rccl_single) and sizes is not None and steps == 2:
        # the five ASPP branches (256 channels each) exchange their SyncBN statistics in ONE message per direction and step
        # (layers/aspp.py _call_grouped): 5 x (2 x 256 + 1, padded to 516 so every layer's slot stays 16-byte aligned) forward,
        # 5 x (2 x 256) backward; the end conv keeps its own pair
        assert sizes.count(5 * 516) == 2 and sizes.count(5 * 512) == 2, sizes
        assert sizes.count(513) == 2 and sizes.count(512) == 2, sizes
    # numpy arrays are pickled by value (torch tensors would travel through /dev/shm handles that die with this process)
    out = {p.iseg_name: p.detach().cpu().numpy().copy() for p in model.parameters()}
    out.update({b.iseg_name: b.detach().cpu().numpy().copy() for b in model.buffers() if hasattr(b, "iseg_name")})
    miou = tm.metric_results()["output_1_IOU"]
    if rccl_single:
        assert tm.reducer.uses, "the reducer never saw a gradient report"
    q.put((rank, world, losses, out, miou))
    if world > 1 or rccl_single:
        dist.barrier()
        torch.distributed.destroy_process_group()


def _run(world, size, batch, rccl_single=False, native=False, graphed=False, steps=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_step, args=(r, world, port, q, size, batch, rccl_single, native, graphed, steps)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    return sorted(res, key=lambda r: r[0])


def test_two_ranks_equal_one_rank_full_batch(cuda):
    size, batch = (64, 64), 4
    single = _run(1, size, batch)[0]
    double = _run(2, size, batch)
    w1 = single[3]
    for rank, world, losses, w, miou in double:
        worst = max((float(abs(w[k] - w1[k]).max()) / max(float(abs(w1[k]).max()), 1e-6), k) for k in w1)
        assert worst[0] < 2e-4, (rank, worst)
        assert abs(miou - single[4]) < 1e-3      # confusion matrices are summed over ranks when the metric is read
    # per-rank losses are per-replica means: their average is the full-batch mean
    for step in range(2):
        avg = sum(d[2][step] for d in double) / 2
        assert abs(avg - single[2][step]) < 1e-4 * max(1.0, abs(single[2][step]))
    for k in double[0][3]:
        assert (double[0][3][k] == double[1][3][k]).all(), f"ranks diverged on {k}"


def test_rccl_backend_single_rank_collectives_are_identity(cuda):
    """the `nccl` (= RCCL) branch of dist.py on real hardware: one rank, but the process group exists and EVERY collective of the step
    is issued -- weight / buffer broadcast, packed SyncBN all-reduces on the compute stream, the asynchronous gradient buckets launched
    from inside backward on RCCL's stream and waited for before the optimizer.  A sum over one rank is the identity, so weights, moving
    statistics and losses after two steps must equal the collective-free run (up to fp32 rounding: the synchronised path normalises ASPP's five branches as a
    group, with other launch shapes for the statistics); a missing stream dependency shows up as a gross difference."""
    size, batch = (64, 64), 4
    plain = _run(1, size, batch)[0]
    rccl = _run(1, size, batch, rccl_single=True)[0]
    for a, b in zip(plain[2], rccl[2]):
        assert abs(a - b) < 1e-5 * max(1.0, abs(a)), (plain[2], rccl[2])
    for k in plain[3]:
        d = float(abs(plain[3][k] - rccl[3][k]).max())
        assert d <= 1e-5 * max(float(abs(plain[3][k]).max()), 1e-3), (k, d)


def test_c_abi_exchange_entry_points_single_rank(cuda):
    """iseg_comm_unique_id / iseg_comm_init / iseg_allreduce_sum / iseg_comm_destroy (include/iseg_hip.h): RCCL through the C ABI on a world of
    one rank -- the sum over one rank is the buffer itself, for the fp32 (SyncBN / gradient) and the bf16 message types, in stream order"""
    import ctypes as C

    from iseg_amd import _hip, kernels as K

    L = _hip.lib()
    uid = (C.c_char * 128)()
    _hip.check(L.iseg_comm_unique_id(uid), "iseg_comm_unique_id")
    assert any(bytes(uid))      # an id was written
    comm = C.c_void_p()
    _hip.check(L.iseg_comm_init(C.byref(comm), 1, 0, uid), "iseg_comm_init")
    assert comm.value
    try:
        g = torch.Generator().manual_seed(0)
        for dtype, code in ((torch.float32, K.F32), (torch.bfloat16, K.BF16)):
            x = torch.randn(100003, generator=g).to(dtype).cuda()
            want = x.clone()
            y = K.axpby(x, None, 2.0, 0.0)      # a kernel of ours in front, on the same stream
            _hip.check(L.iseg_allreduce_sum(comm, K.ptr(y), y.numel(), code, K.stream()), "iseg_allreduce_sum")
            z = K.axpby(y, None, 0.5, 0.0)      # ... and one behind
            torch.cuda.synchronize()
            assert torch.equal(z, want)
        assert L.iseg_allreduce_sum(comm, None, 0, K.F32, K.stream()) == 0      # empty message
        assert L.iseg_allreduce_sum(comm, K.ptr(x), 8, 7, K.stream()) != 0      # unknown dtype is refused
    finally:
        _hip.check(L.iseg_comm_destroy(comm), "iseg_comm_destroy")
    bad = C.c_void_p()
    assert L.iseg_comm_init(C.byref(bad), 2, 2, uid) != 0      # rank outside the world


def test_stream_ordered_exchange_single_rank_eager_and_replayed(cuda):
    """ISEG_DIST_NATIVE=1 (round 4): SyncBN messages and gradient buckets go through iseg_allreduce_sum -- the C ABI's own RCCL communicator --
    as plain stream-ordered enqueues (buckets on a side stream forked / joined by events) instead of c10d work objects.  On a world of one
    rank every sum is the identity, so (1) the eager step equals the c10d-exchange step BIT FOR BIT (same kernels, same order: the step is
    deterministic) and (2) the same step captured into ONE HIP graph -- collectives included, the point of the exercise: c10d's asynchronous
    work objects cannot be captured -- replays bit for bit what the eager steps produce."""
    size, batch = (64, 64), 4
    c10d = _run(1, size, batch, rccl_single=True, steps=4)[0]
    eager = _run(1, size, batch, rccl_single=True, native=True, steps=4)[0]
    replay = _run(1, size, batch, rccl_single=True, native=True, graphed=True, steps=4)[0]
    assert c10d[2] == eager[2] == replay[2], (c10d[2], eager[2], replay[2])
    for k in c10d[3]:
        assert (c10d[3][k] == eager[3][k]).all(), f"stream-ordered exchange differs from the c10d exchange on {k}"
        assert (eager[3][k] == replay[3][k]).all(), f"the replayed data-parallel step differs from the eager one on {k}"
