"""World-size-2 data-parallel tests on CPU (gloo): the arena / bucketed gradient exchange that bench.py and train.py run over
RCCL on the GPUs.  Device-agnostic plumbing only -- no HIP kernel is involved, nothing here touches
the oracle."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_params(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(16, 17, 3, 3), (16,), (1, 16, 3, 3), (1,), (32, 16, 4, 4), (16,), (64, 3, 3, 3), (64,), (64,), (64,), (7,)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]


def _worker_reduce(rank, world, port, bucket_bytes, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from supervised_dispnet_amd.distributed import GradReducer, shard_slice
        from supervised_dispnet_amd.optim import ParamArena
        params = _make_params()
        order = list(reversed(params))                       # gradients are produced last-layer-first
        arena = ParamArena(params, production_order=order)
        assert [id(p) for p in arena.params] == [id(p) for p in order]
        red = GradReducer(arena, bucket_bytes=bucket_bytes)
        # buckets tile the arena contiguously, in production order
        assert red.buckets[0]["lo"] == 0 and red.buckets[-1]["hi"] == arena.numel
        for a, b in zip(red.buckets[:-1], red.buckets[1:]):
            assert a["hi"] == b["lo"]
        for step in range(2):                                # two steps: the reducer must re-arm itself
            for i, p in enumerate(arena.params):
                p._dn_grad_view.fill_(float((rank + 1) * (i + 1) + step))
                if i != 3:                                   # one parameter never reports (e.g. no gradient this step)
                    red.grad_ready(p)
            scale = red.finish()
            assert scale == 1.0 / world
            for i, p in enumerate(arena.params):
                want = sum((r + 1) * (i + 1) + step for r in range(world))
                assert torch.all(p._dn_grad_view == want), (step, i)
            # padding slots between 16-byte aligned slices stay zero
            used = torch.zeros(arena.numel, dtype=torch.bool)
            for p, o in zip(arena.params, arena.offsets):
                used[o:o + p.numel()] = True
            assert torch.all(arena.flat_g[~used] == 0)
        # the batch shards exactly like DataParallel.scatter: contiguous slices in rank order
        sl = shard_slice(32, rank, world)
        assert (sl.start, sl.stop) == (rank * 16, rank * 16 + 16)
        out.put((rank, "ok"))
    except Exception as e:      # surface the failure in the parent
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [1 << 10, 1 << 30])
def test_bucketed_gradient_allreduce_world2(bucket_bytes):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_reduce, args=(r, 2, port, bucket_bytes, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _worker_loss_counts(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from supervised_dispnet_amd.distributed import global_masked_mean
        # whole-batch masked means (Multiscale_* / DORN normalise by the GLOBAL valid count, loss_functions.py:232-237,72-73):
        # rank-local (sum, count) pairs are exchanged before the division
        local_sum = torch.tensor([3.0 if rank == 0 else 10.0])
        local_cnt = torch.tensor([2.0 if rank == 0 else 6.0])
        m = global_masked_mean(local_sum, local_cnt)
        assert torch.allclose(m, torch.tensor([13.0 / 8.0]))
        out.put((rank, "ok"))
    except Exception as e:
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_global_masked_mean_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_loss_counts, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_reducer_is_a_no_op():
    from supervised_dispnet_amd.distributed import GradReducer, shard_slice
    from supervised_dispnet_amd.optim import ParamArena
    arena = ParamArena(_make_params())
    red = GradReducer(arena)
    for p in arena.params:
        p._dn_grad_view.fill_(2.0)
        red.grad_ready(p)
    assert red.finish() == 1.0
    assert torch.all(arena.params[0]._dn_grad_view == 2.0)
    assert shard_slice(32, 0, 1) == slice(0, 32)
    with pytest.raises(ValueError):
        shard_slice(30, 0, 4)


def _worker_stats_and_plain(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from supervised_dispnet_amd.distributed import average_plain_grads, data_parallel_world, exchange_loss_stats
        assert data_parallel_world() == world
        # per-group statistics [G, 8]: columns 0..2 are sums, column 3 is a maximum, the rest must be left alone
        st = torch.arange(16, dtype=torch.float32).view(2, 8) * (rank + 1)
        keep = st.clone()
        exchange_loss_stats(st, sum_cols=(0, 1, 2), max_cols=(3,))
        tot = sum(r + 1 for r in range(world))
        base = torch.arange(16, dtype=torch.float32).view(2, 8)
        assert torch.equal(st[:, :3], base[:, :3] * tot)
        assert torch.equal(st[:, 3], base[:, 3] * world)              # max over ranks of base * (rank + 1)
        assert torch.equal(st[:, 4:], keep[:, 4:])
        # whole-batch masked mean = sum of sums / sum of counts: identical to the single-process value on the gathered batch
        g = torch.Generator().manual_seed(5)
        gt = torch.rand(4, 6, 8, generator=g) * 100.0 - 10.0          # some values outside (0, 80)
        pred = torch.rand(4, 6, 8, generator=g) * 80.0
        valid = (gt > 0) & (gt < 80)
        want = (gt[valid] - pred[valid]).abs().mean()
        sl = slice(rank * 2, rank * 2 + 2)
        v = valid[sl]
        loc = torch.zeros(1, 8)
        loc[0, 0] = (gt[sl][v] - pred[sl][v]).abs().sum()
        loc[0, 1] = v.sum()
        exchange_loss_stats(loc, sum_cols=(0, 1, 2), max_cols=(3,))
        assert torch.allclose(loc[0, 0] / loc[0, 1], want, rtol=1e-6)
        # --sgd / --diff-lr path: .grad tensors are averaged over the ranks; a parameter without gradient stays without
        ps = _make_params()
        for i, p in enumerate(ps):
            if i != 2:
                p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
        average_plain_grads(ps)
        for i, p in enumerate(ps):
            if i == 2:
                assert p.grad is None
            else:
                assert torch.allclose(p.grad, torch.full_like(p, (i + 1) * tot / world))
        out.put((rank, "ok"))
    except Exception as e:
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_loss_stats_exchange_and_plain_grad_average_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_stats_and_plain, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


@pytest.mark.parametrize("n,gb,world", [(1000, 32, 8), (37, 8, 4), (64, 32, 8), (5, 8, 2), (33, 32, 8)])
def test_rank_sampler_covers_every_sample_once_and_never_yields_an_empty_batch(n, gb, world):
    """drop_last=False (the validation loader, train.py): the partial tail batch must not hand an empty index list to any rank
    (default_collate([]) raises) and the union over ranks must be every sample exactly once."""
    import torch.utils.data as tud
    from supervised_dispnet_amd.data import RankSampler

    class Range(tud.Dataset):
        def __len__(self):
            return n

        def __getitem__(self, i):
            return torch.tensor(i)

    seen = []
    for rank in range(world):
        s = RankSampler(n, gb, rank, world, shuffle=False, drop_last=False)
        batches = list(tud.DataLoader(Range(), batch_sampler=s))
        assert len(batches) == len(s)
        assert all(b.numel() > 0 for b in batches)
        seen += [int(v) for b in batches for v in b]
    assert sorted(seen) == list(range(n))
    # training sampler (drop_last=True): equal slices on every rank, tail dropped
    lens = set()
    for rank in range(world):
        s = RankSampler(n, gb, rank, world, shuffle=True, seed=3, drop_last=True)
        bs = list(s)
        assert len(bs) == len(s) == n // gb and all(len(b) == gb // world for b in bs)
        lens.add(len(bs))
    assert len(lens) == 1


def test_bench_dry_run_two_ranks_constructs_buckets_and_tears_down():
    """bench.py --gpus 2 --dry-run under gloo on CPU tensors: arena in gradient-production order, bucketing, one all-reduce
    cycle through the GradSink hooks the engine uses, strong-scaling shard, process-group tear-down."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--scaling", "strong"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["ok"] and d["world"] == 2 and d["buckets"] >= 3 and d["shard"] == [0, 16] and d["arena_params"] == 80


def test_bench_spawns_its_own_ranks_when_invoked_plainly():
    """`python bench.py --gpus 2 --dry-run` with NO launcher and no WORLD_SIZE in the environment (the reference's train.py needs no
    launcher either, train.py:316-317): bench.py re-executes itself as two ranks under torch.distributed.run on a free port."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["ok"] and d["world"] == 2 and d["shard"] == [0, 16] and d["arena_params"] == 80


def _worker_two_communicators(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        from supervised_dispnet_amd import distributed as DD
        from supervised_dispnet_amd.optim import ParamArena
        # The gradient buckets travel on ONE communicator (here: a second process group standing in for the library's own RCCL
        # communicator), the whole-batch loss statistics on ANOTHER (the default group), interleaved every step: forward statistics,
        # then the backward's buckets as their gradients land.  Safe only if every rank issues them in the same order -- the ranks
        # run at different speeds on purpose.
        grads_group = dist.new_group(ranks=list(range(world)), backend="gloo")
        arena = ParamArena(_make_params(), production_order=None)
        red = DD.GradReducer(arena, bucket_bytes=1 << 10, process_group=grads_group, comm="torch")
        DD.ISSUE_LOG = []
        for step in range(3):
            time.sleep(0.05 * rank * (step + 1))                       # rank skew
            st = torch.full((4, 8), float(rank + 1 + step))
            DD.exchange_loss_stats(st, sum_cols=(0, 1, 2), max_cols=(3,))             # "forward": default group
            assert torch.all(st[:, 0] == sum(r + 1 + step for r in range(world))) and torch.all(st[:, 3] == world + step)
            for i, p in enumerate(arena.params):                       # "backward": buckets launch as gradients land
                p._dn_grad_view.fill_(float((rank + 1) * (i + 1)))
                if (i + rank) % 3 == 0:
                    time.sleep(0.01)
                red.grad_ready(p)
            red.finish()
            for i, p in enumerate(arena.params):
                assert torch.all(p._dn_grad_view == sum((r + 1) * (i + 1) for r in range(world)))
        logs = [None] * world
        dist.all_gather_object(logs, DD.ISSUE_LOG)
        assert all(l == logs[0] for l in logs), "ranks issued their collectives in different orders"
        kinds = [e[0] for e in logs[0]]
        per_step = len(kinds) // 3
        assert kinds[:2] == ["stats", "stats"] and set(kinds[2:per_step]) == {"bucket"} and per_step >= 5
        out.put((rank, "ok"))
    except Exception as e:
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_interleaved_communicators_keep_one_issue_order_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_two_communicators, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _worker_comm_agreement(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from supervised_dispnet_amd import distributed as DD
        # a decision every rank must take together: one dissenting rank turns it down for all
        assert DD.agree_all_ranks(True) is True
        assert DD.agree_all_ranks(rank != 1) is False
        # CPU arenas never take the own-RCCL path; the reducer names the path that runs
        from supervised_dispnet_amd.optim import ParamArena
        red = DD.GradReducer(ParamArena(_make_params()))
        assert red.comm is None and red.path == "torch.distributed:gloo" and red.world == world
        out.put((rank, "ok"))
    except Exception as e:
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_comm_choice_is_collective_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_comm_agreement, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_last_gradient_bucket_is_cut_short():
    """GradReducer: contiguous buckets covering the arena exactly once, in arena (= production) order, each at least bucket_bytes except
    the last one, which is only the final ~tail_bytes of the arena -- the one all-reduce nothing can overlap starts when the first
    layer's gradient lands."""
    import torch
    from supervised_dispnet_amd.distributed import GradReducer

    class Arena(object):
        pass

    sizes = [1, 1152, 16, 4608, 32, 27648, 64, 110592, 128, 442368, 1769472, 2097152, 512, 2359296, 512, 2359296, 2359296, 1179648,
             589824, 589824, 294912, 147456, 73728, 36864, 1728]
    a = Arena()
    a.params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    a.offsets, t = [], 0
    for n in sizes:
        a.offsets.append(t)
        t += (n + 3) // 4 * 4
    a.numel, a.flat_g = t, torch.zeros(t)
    for tail in (1 << 20, 0):
        r = GradReducer(a, bucket_bytes=8 << 20, tail_bytes=tail)
        b = r.buckets
        assert b[0]["lo"] == 0 and b[-1]["hi"] == t and all(b[i]["hi"] == b[i + 1]["lo"] for i in range(len(b) - 1))
        assert sum(len(x["params"]) for x in b) == len(sizes)
        assert all((x["hi"] - x["lo"]) * 4 >= (8 << 20) for x in b[:-2])          # (the bucket in front of the tail is whatever is left)
        if tail:
            assert (1 << 20) <= (b[-1]["hi"] - b[-1]["lo"]) * 4 <= (4 << 20)          # ~1 MB: until the next parameter boundary
    assert len(GradReducer(a, bucket_bytes=8 << 20, tail_bytes=1 << 20).buckets) >= len(GradReducer(a, bucket_bytes=8 << 20, tail_bytes=0).buckets)
