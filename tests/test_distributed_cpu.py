"""The N>1 path on CPU: world_size-2 (and 3, uneven shards) gloo processes run the
sharded sampler with the CPU oracle injected as compute backend, and must
reproduce the single-process result exactly (rows are independent given the
shared MC set, so sharding may not change a single bit)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleBackend:
    """Test-only backend: routes the three compute calls to the CPU oracle."""

    def __init__(self):
        from oracle import oracle as O
        from helpers import oracle_net
        self.O = O
        self.nets = {"x": oracle_net("mnist32"), "y": oracle_net("svhn"), "r": oracle_net("ratio_ms")}

    def _net(self, model):
        return self.nets["x"] if model.in_channels == 1 else self.nets["y"]

    def sample_single(self, model, x, num_steps):
        d, b = self._net(model)
        x.copy_(torch.from_numpy(self.O.sample_single(d, b, x.numpy(), num_steps)))
        return x

    def ratios(self, ratio_estimator, mc_x1, mc_y1):
        kind, b = self.nets["r"]
        return torch.from_numpy(self.O.ratio_eval(kind, b, mc_x1.numpy(), mc_y1.numpy(), "ratio", "disc"))

    def sample_pair(self, fm_x, fm_y, x, y, mc_x1, mc_y1, mc_r, num_steps, gamma):
        (dx, bx), (dy, by) = self.nets["x"], self.nets["y"]
        n = lambda t: None if t is None else t.numpy()
        ox, oy = self.O.sample_pair(dx, bx, dy, by, x.numpy(), y.numpy(), n(mc_x1), n(mc_y1), n(mc_r), num_steps, gamma)
        x.copy_(torch.from_numpy(ox))
        y.copy_(torch.from_numpy(oy))
        return x, y


B, N, S, GAMMA, SEED = 5, 3, 2, 0.5, 123


def _run(rank, world, port, guided, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import make_module, paired_noise
    from ratio_guided_multimodal_fm_amd.distributed import sharded_paired_sampler
    fm, fs, rr = make_module("mnist32"), make_module("svhn"), make_module("ratio_ms")
    noise = paired_noise(SEED, B, N, (1, 32, 32), (3, 32, 32))
    for gather in ("rank0", "all"):
        x, y = sharded_paired_sampler(fm, fs, rr if guided else None, "mc_feng" if guided else "none", GAMMA, S,
                                      noise, torch.device("cpu"), backend=OracleBackend(), gather=gather)
        if gather == "rank0":
            assert (x is None) == (rank != 0)
        if x is not None:
            np.savez(os.path.join(out_dir, f"out_{gather}_{rank}.npz"), x=x.numpy(), y=y.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,guided", [(2, True), (3, True), (2, False)])
def test_sharded_equals_single_process(tmp_path, world, guided):
    from oracle import oracle as O
    from helpers import oracle_net, paired_noise
    port = 29600 + world * 7 + int(guided)
    mp.spawn(_run, args=(world, port, guided, str(tmp_path)), nprocs=world, join=True)
    (dx, bx), (dy, by), (kind, br) = oracle_net("mnist32"), oracle_net("svhn"), oracle_net("ratio_ms")
    noise = tuple(v.numpy() for v in paired_noise(SEED, B, N, (1, 32, 32), (3, 32, 32)))
    rx, ry, _ = O.paired_sampler(dx, bx, dy, by, kind, br, "disc", noise, guided, GAMMA, S)
    got = np.load(tmp_path / "out_rank0_0.npz")
    assert np.array_equal(got["x"], rx) and np.array_equal(got["y"], ry)
    for r in range(world):
        g = np.load(tmp_path / f"out_all_{r}.npz")
        assert np.array_equal(g["x"], rx) and np.array_equal(g["y"], ry)


def _run_sweep(rank, world, port, out_dir):
    """The multi-GPU launcher's data path (evaluate_mnist_svhn --sharded) on gloo ranks: run_sweep with the sharded
    sampler, every rank seeded alike, one generator stream across the configurations."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    import json
    from helpers import make_module
    from ratio_guided_multimodal_fm_amd.distributed import init_from_env, make_sharded_sampler
    from ratio_guided_multimodal_fm_amd.evaluate_mnist_svhn import run_sweep
    from ratio_guided_multimodal_fm_amd.utils import set_seed
    r, w, device = init_from_env("gloo")
    assert (r, w) == (rank, world)
    fm, fs, rr = make_module("mnist32"), make_module("svhn"), make_module("ratio_ms")
    cm, cs = make_module("clf_mnist"), make_module("clf_svhn")
    produced = []
    inner = make_sharded_sampler((1, 32, 32), (3, 32, 32), backend=OracleBackend(), gather="all")

    def sampler(*a):
        out = inner(*a)
        produced.append(out)
        return out

    set_seed(77)
    res = run_sweep(fm, fs, lambda: rr, cm, cs, ["none", "mc_feng"], [0.0, 2.0], B, S, device, N, sampler=sampler)
    np.savez(os.path.join(out_dir, f"sweep_{rank}.npz"), **{f"x{i}": p[0].numpy() for i, p in enumerate(produced)},
             **{f"y{i}": p[1].numpy() for i, p in enumerate(produced)})
    with open(os.path.join(out_dir, f"sweep_{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


def test_sharded_sweep_launcher_path(tmp_path):
    import json
    from oracle import oracle as O
    from helpers import oracle_net
    world = 2
    mp.spawn(_run_sweep, args=(world, 29655, str(tmp_path)), nprocs=world, join=True)
    (dx, bx), (dy, by), (kind, br) = oracle_net("mnist32"), oracle_net("svhn"), oracle_net("ratio_ms")
    # single-process replay of the same generator stream: configurations (none, 0), (mc_feng, 0), (mc_feng, 2)
    torch.manual_seed(77)
    want = []
    for guided, gamma in ((False, 0.0), (True, 0.0), (True, 2.0)):
        x0, y0 = torch.randn(B, 1, 32, 32), torch.randn(B, 3, 32, 32)
        mx = torch.randn(N, 1, 32, 32) if guided else None
        my = torch.randn(N, 3, 32, 32) if guided else None
        noise = tuple(None if v is None else v.numpy() for v in (x0, y0, mx, my))
        want.append(O.paired_sampler(dx, bx, dy, by, kind, br, "disc", noise, guided, gamma, S)[:2])
    res = [json.load(open(tmp_path / f"sweep_{r}.json")) for r in range(world)]
    assert res[0] == res[1] and [(r["method"], r["guidance_strength"]) for r in res[0]] == [("none", 0.0), ("mc_feng", 0.0), ("mc_feng", 2.0)]
    for r in range(world):
        g = np.load(tmp_path / f"sweep_{r}.npz")
        for i, (wx, wy) in enumerate(want):
            assert np.array_equal(g[f"x{i}"], wx) and np.array_equal(g[f"y{i}"], wy)


def test_shard_bounds_cover_everything():
    from ratio_guided_multimodal_fm_amd.distributed import shard_bounds
    for n in (0, 1, 7, 256, 4096, 8193):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_grad_log_ratio_is_routed_or_refused():
    """`--sharded --guidance_method grad_log_ratio` must not silently produce unguided samples.  The sharded
    sampler routes it to the backend's sample_pair_grad on each rank's rows (no MC set is drawn or integrated), and a
    backend without that call -- or an unknown method -- raises."""
    import torch
    from ratio_guided_multimodal_fm_amd.distributed import sharded_paired_sampler

    class Dummy(torch.nn.Module):
        def forward(self, *a):
            raise AssertionError("not evaluated")

    calls = []

    class Backend:
        def sample_single(self, *a):
            calls.append("single")

        def ratios(self, *a):
            calls.append("ratios")

        def sample_pair(self, fm_x, fm_y, x, y, mx, my, mr, steps, gamma):
            calls.append(("pair", mx is None))

        def sample_pair_grad(self, fm_x, fm_y, ratio, x, y, steps, gamma):
            calls.append(("grad", tuple(x.shape), steps, gamma))
            x.add_(1.0)

    fx, fy, rr = Dummy(), Dummy(), Dummy()
    x0, y0 = torch.zeros(5, 1, 4, 4), torch.zeros(5, 3, 4, 4)
    xs, ys = sharded_paired_sampler(fx, fy, rr, 'grad_log_ratio', 0.7, 9, (x0, y0, None, None), torch.device('cpu'),
                                    backend=Backend())
    assert calls == [("grad", (5, 1, 4, 4), 9, 0.7)] and float(xs.min()) == 1.0 and float(x0.max()) == 0.0

    class NoGrad:
        sample_single, ratios, sample_pair = Backend.sample_single, Backend.ratios, Backend.sample_pair

    nb = NoGrad()
    with pytest.raises(ValueError):
        sharded_paired_sampler(fx, fy, rr, 'grad_log_ratio', 0.7, 9, (x0, y0, None, None), torch.device('cpu'), backend=nb)
    with pytest.raises(ValueError):
        sharded_paired_sampler(fx, fy, rr, 'no_such_method', 0.7, 9, (x0, y0, None, None), torch.device('cpu'), backend=Backend())
    # without an estimator the reference's samplers fall back to unguided integration (sample_mnist_svhn.py:85,124)
    calls.clear()
    sharded_paired_sampler(fx, fy, None, 'grad_log_ratio', 0.7, 9, (x0, y0, None, None), torch.device('cpu'), backend=Backend())
    assert calls == [("pair", True)]


def test_hsa_ipc_mode_is_set_when_the_launchers_start(monkeypatch):
    """the dmabuf-IPC setting must be in the environment before the first HIP call of a launcher process:
    importing the distributed module (the first statement of both launchers' main()) puts it there."""
    import importlib
    import os
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    import ratio_guided_multimodal_fm_amd.distributed as d
    importlib.reload(d)
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    import inspect
    from ratio_guided_multimodal_fm_amd import evaluate_mnist_svhn, sample_mnist_svhn
    for mod in (sample_mnist_svhn, evaluate_mnist_svhn):
        body = inspect.getsource(mod.main).split("\n")
        assert "from . import distributed" in body[1], body[1]
