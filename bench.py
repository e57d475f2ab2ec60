#!/usr/bin/env python3
"""Headline benchmark: paired images/sec of the ratio-guided sampler
(MNIST32 1x32x32 + SVHN 3x32x32, mc_feng guidance 0.5, 100 Euler steps,
N_mc = 256, batch 512 PER GPU) -- BASELINE.json configs[2] at 1 GPU, configs[3]
at 8 GPUs (weak scaling, 512 rows per rank).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one whole sampling call: sharded MC pre-phase + ratio estimator +
all_gather of the MC set + the guided Euler loop + the gather of the outputs,
on noise that is already resident in HBM.  Synthetic parameters
(ratio_guided_multimodal_fm_amd/synth.py, no trained checkpoint exists) and
CPU-generator noise seeded 42 (the reference CLI's default seed).

Rank 0 prints ONE JSON line with the driver's contract fields plus
  roofline     : the conv kernel class (the dominant kernels, matrix-core-bound), hipEvent-timed on the
                 launch streams inside the timed region (librgfm_hip's rgfm_profile_*).  Default arithmetic
                 (RGFM_CONV unset / hx2): fp32 operands as two scaled fp16 planes, three f16-MFMA products per
                 fp32 product, fp32 accumulate -> peak = dense f16 MFMA peak / 3 in fp32-equivalent TFLOP/s
                 (`peak` = `peak_nominal`: the guide's 2500 TFLOP/s; `peak_sustained`: the rate a register-only
                 MFMA loop holds on THIS device under DVFS, measured in this process after the timed region).
                 RGFM_CONV=bx3: three bf16 planes, six products (peak / 6); RGFM_CONV=f32: v_mfma_f32_32x32x2_f32,
                 peak 157.3.  At N=1 one extra (untimed-for-`value`) call in exact-fp32 mode is reported as
                 roofline.exact_fp32_mode.  roofline.hbm_kernels: the HBM-bound kernels of a step (guidance,
                 first / last conv of each net) with their algorithmic GB/s against the nominal 8 TB/s and
                 against a float4 copy measured in this process.
  cpu_baseline : the CPU oracle (a C port of the reference algorithm, test
                 infrastructure) timed on this host's cores on a bounded sample (N=1 only).
  parity_check : BASELINE.md section 4's gate on the number itself: eight rows of the LAST timed call are followed by
                 the CPU oracle over all Euler steps from that call's own (x0, y0) and the MC set it produced;
                 max |difference| must be within 1e-4 or the run fails (rank 0, after the timed region).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / f16 MFMA
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured with a float4 copy)
ARITHMETIC = {
    "hx2": "f32 emulated on the f16 matrix cores: operands as 2 scaled fp16 planes (22-bit significands), 3 of the 4 plane "
           "products per fp32 product, fp32 accumulate; range-guarded on both sides: a call whose activations leave the window is repeated on split-bf16 (high side) or on the fp32 MFMA (low side) (DESIGN.md sections 2 and 4)",
    "bx3": "f32 emulated on the bf16 matrix cores: operands as 3 exact bf16 planes (24-bit significands), 6 products per "
           "fp32 product, fp32 accumulate",
    "f32": "f32 on the matrix cores: v_mfma_f32_32x32x2_f32",
}
PRODUCTS = {"hx2": 3, "bx3": 6}  # 16-bit MFMA products per fp32 multiply-add (conv_mfma_hx2*.hip, conv_mfma_bx3.hip)
KCLASS = {"conv": 0, "other": 1, "conv_in<1>": 2, "conv_in<3>": 3, "conv_out<1>": 4, "conv_out<3>": 5,
          "guid_logp": 6, "guid_apply": 7}  # include/rgfm.h RGFM_KCLASS_*


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch-per-gpu", type=int, default=512)
    p.add_argument("--mc", type=int, default=256)
    p.add_argument("--euler-steps", type=int, default=100)
    p.add_argument("--gamma", type=float, default=0.5)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-timers", action="store_true",
                   help="skip the hipEvent kernel-class timers (use under rocprofv3)")
    p.add_argument("--no-alt-mode", action="store_true",
                   help="skip the extra call in exact-fp32-MFMA mode (reported beside the default mode)")
    p.add_argument("--no-parity-check", action="store_true",
                   help="skip the oracle check of the last timed call's rows (use under rocprofv3)")
    p.add_argument("--no-arith-check", action="store_true",
                   help="skip the small float64 error check after the timed region (use under rocprofv3: the trace "
                        "then ends with the guided main loop)")
    return p.parse_args()


def cpu_baseline(fm, fs, rr, euler_steps, B_full, N_full):
    """Oracle timing on a bounded sample of the same workload (reported, not the target)."""
    from oracle import oracle as O
    from ratio_guided_multimodal_fm_amd.synth import paired_noise
    cores = O.num_threads()
    # one row per core (the oracle threads over rows), the workload's own N_mc; the number
    # of Euler steps is what bounds the sample to ~10-30 s
    B, N, S = max(16, cores), N_full, 6
    noise = tuple(v.numpy() for v in paired_noise(42, B, N, (1, 32, 32), (3, 32, 32)))
    dx, bx, dy, by, br = O.desc_of(fm), O.blob_of(fm), O.desc_of(fs), O.blob_of(fs), O.blob_of(rr)
    t0 = time.perf_counter()
    # first S of `euler_steps` Euler steps of the MC pre-phase, the ratio net, and the guided loop
    mx = O.sample_single(dx, bx, noise[2], euler_steps, 0, S)
    my = O.sample_single(dy, by, noise[3], euler_steps, 0, S)
    r = O.ratio_eval("mnist_svhn", br, mx, my, "ratio", "disc")
    O.sample_pair(dx, bx, dy, by, noise[0], noise[1], mx, my, r, euler_steps, 0.5, 0, S)
    dt = time.perf_counter() - t0
    row_steps = (B + N) * S                      # U-Net pair evaluations done (>99.9 % of the work)
    full = (B_full + N_full) * euler_steps       # pair evaluations of one full call
    return {
        "value": B_full / (dt * full / row_steps),
        "unit": "paired images/sec",
        "cores": O.num_threads(),
        "kind": "port",
        "sample": f"B={B}, N_mc={N}, {S} of {euler_steps} Euler steps (pre-phase + ratio + guided loop) "
                  f"= {row_steps} U-Net pair evaluations in {dt:.1f} s, scaled to the {full} of one full call",
    }


PARITY_TOL = 1e-4  # BASELINE.md section 4 / SURVEY 8c: max |difference| of a full sampler call on outputs in [-6, 6]


def parity_check(fm, fs, x0, y0, kept, out, euler_steps, gamma, nrows=8):
    """Rows of the timed call against the CPU oracle: same (x0, y0), the MC set THAT call produced, every Euler step."""
    import numpy as np
    from oracle import oracle as O
    B = x0.shape[0]
    rows = sorted(set(int(round(i * (B - 1) / max(nrows - 1, 1))) for i in range(nrows)))
    t0 = time.perf_counter()
    ox, oy = O.sample_pair(O.desc_of(fm), O.blob_of(fm), O.desc_of(fs), O.blob_of(fs), x0[rows].cpu().numpy(),
                           y0[rows].cpu().numpy(), kept["mc_x1"].cpu().numpy(), kept["mc_y1"].cpu().numpy(),
                           kept["mc_ratios"].cpu().numpy(), euler_steps, gamma, 0, euler_steps)
    dx = float(np.abs(out[0][rows].cpu().numpy() - ox).max())
    dy = float(np.abs(out[1][rows].cpu().numpy() - oy).max())
    return {"rows": len(rows), "row_indices": rows, "max_abs": max(dx, dy), "max_abs_x": dx, "max_abs_y": dy,
            "tolerance": PARITY_TOL, "steps": euler_steps,
            "what": "rows of the LAST timed call (its x0, y0 and the MC set it produced) followed by the CPU oracle over "
                    "all Euler steps", "oracle_seconds": time.perf_counter() - t0}


def kernel_sources_sha():
    """sha256 over the conv kernel sources: ties a committed PMC traffic file to the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "ratio_guided_multimodal_fm_amd", "csrc")
    for name in ("conv_hx2_common.h", "conv_mfma_hx2.hip", "conv_mfma_hx2p.hip", "conv_mfma_hx2q.hip", "conv_mfma_hx2s.hip", "conv_mfma_hx2c.hip", "conv_mfma_hx2d.hip", "conv_mfma_bx3.hip",
                 "conv_mfma.hip", "rgfm_device.h", "rgfm_kernels.h"):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under a torch.distributed.run launch (RANK in the environment) the RCCL group is made at ANY world size, so a
    # one-rank launch walks the same barrier / gather / all_reduce code as the driver's N > 1 runs
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from ratio_guided_multimodal_fm_amd import _engine, _lib, models as M
    from ratio_guided_multimodal_fm_amd.distributed import sharded_paired_sampler
    from ratio_guided_multimodal_fm_amd.synth import load_synth, paired_noise

    fm = load_synth(M.FlowMatchingUNetMNIST(32), 0).eval()
    fs = load_synth(M.FlowMatchingUNetSVHN(), 1).eval()
    rr = load_synth(M.RatioEstimatorMNISTSVHN(), 2).eval()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(fm, fs, rr, args.euler_steps, args.batch_per_gpu, args.mc)
    fm, fs, rr = fm.to(dev), fs.to(dev), rr.to(dev)

    B = args.batch_per_gpu * world
    # host noise (reference draw order), identical on every rank; one device-resident copy per call
    x0, y0, mx0, my0 = paired_noise(42, B, args.mc, (1, 32, 32), (3, 32, 32))
    calls = args.warmup + args.steps
    resident = [tuple(t.to(dev) for t in (x0, y0, mx0, my0)) for _ in range(calls)]
    torch.cuda.synchronize()

    kept = {}  # references to the MC set of the latest call (no copies, no synchronisation): the parity gate reads them

    def one_call(i):
        return sharded_paired_sampler(fm, fs, rr, "mc_feng", args.gamma, args.euler_steps, resident[i], dev,
                                      gather="rank0", keep=kept)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    timers = not args.no_kernel_timers
    if args.warmup == 0 and timers:
        raise SystemExit("the kernel timers need one warm-up call to size their event pool: use --warmup >= 1")
    for i in range(args.warmup):
        if timers and i == args.warmup - 1:
            _engine.profile(enable=True, reset=True)  # count the launches of one call ...
        one_call(i)
    fence()
    if timers:
        per_call = sum(_engine.profile_read(k)[2] for k in KCLASS.values())
        _engine.profile(enable=False)
        # ... and create every hipEvent the timed region will record BEFORE it starts
        _engine.profile(reserve=int(per_call * (args.steps + 2) * 1.05) + 64)
        _engine.profile(enable=True, reset=True)
    t0 = time.perf_counter()
    for i in range(args.warmup, calls):
        out = one_call(i)
    fence()
    elapsed = time.perf_counter() - t0
    conv_ms = conv_sum_ms = conv_n = conv_fl = 0.0
    hbm_classes = {}
    if timers:
        conv_ms, conv_sum_ms, conv_n, conv_fl = _engine.profile_read(0)
        for name, k in KCLASS.items():
            if k >= 2:
                hbm_classes[name] = _engine.profile_read(k)
        _engine.profile(enable=False)
    ranks = None
    if use_dist:
        # what every rank saw, gathered over RCCL: its rank, the world size of ITS process group, its device ordinal and its
        # own time for the K calls -- the line then proves that N ranks on N devices ran
        mine = torch.tensor([float(rank), float(dist.get_world_size()), float(torch.cuda.current_device()), elapsed],
                            device=dev, dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ranks = [{"rank": int(v[0]), "world_size_seen": int(v[1]), "device": int(v[2]), "seconds": float(v[3])}
                 for v in (t.cpu().tolist() for t in allr)]
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    parity = None
    if rank == 0 and not args.no_parity_check:
        # the number is reported only with the outputs of the timed run inside the tolerance
        parity = parity_check(fm, fs, resident[calls - 1][0], resident[calls - 1][1], dict(kept), out, args.euler_steps,
                              args.gamma)
    conv_mode_env = os.environ.get("RGFM_CONV")
    conv_mode = conv_mode_env or "hx2"
    alt = None
    if rank == 0 and world == 1 and timers and conv_mode != "f32" and not args.no_alt_mode:
        # the same call once more with the exact-fp32 MFMA conv (the switch is read per launch)
        os.environ["RGFM_CONV"] = "f32"
        one_call(calls - 1)
        fence()
        runs = []
        for _ in range(3):
            _engine.profile(enable=True, reset=True)
            ta = time.perf_counter()
            one_call(calls - 1)
            fence()
            el = time.perf_counter() - ta
            a_ms, _, a_n, a_fl = _engine.profile_read(0)
            _engine.profile(enable=False)
            runs.append((el, a_ms, a_fl))
        runs.sort()
        alt_elapsed, a_ms, a_fl = runs[1]  # the median call
        if conv_mode_env is None:
            os.environ.pop("RGFM_CONV", None)
        else:
            os.environ["RGFM_CONV"] = conv_mode_env
        ach_a = a_fl / (a_ms * 1e-3) / 1e12
        alt = {"conv": "v_mfma_f32_32x32x2_f32 (RGFM_CONV=f32)", "value": B / alt_elapsed, "unit": "paired images/sec",
               "achieved": ach_a, "peak": PEAK_FP32_MFMA_TFLOPS, "frac": ach_a / PEAK_FP32_MFMA_TFLOPS, "calls": 3,
               "statistic": "median call"}

    arith = None
    gpath = os.path.join(ROOT, "tests", "golden", "fp64_eval.npz")
    if rank == 0 and world == 1 and os.path.exists(gpath) and not args.no_arith_check:
        # error of one SVHN-net evaluation against the reference evaluated in float64 (committed golden vector,
        # tests/golden/make_golden.py) in every conv arithmetic mode: the split-operand paths are fp32-class
        import numpy as np
        from ratio_guided_multimodal_fm_amd.synth import load_synth as _ls
        g = np.load(gpath)
        net = _ls(M.FlowMatchingUNetSVHN(), 14).eval().to(dev)  # the weights the golden vector was made with
        xg = torch.randn(4, 3, 32, 32, generator=torch.Generator().manual_seed(91)).to(dev)
        tg = torch.tensor([0.05, 0.37, 0.71, 0.99], device=dev)
        errs = {}
        fb0 = _engine.range_fallbacks
        for mode in ("hx2", "bx3", "f32"):
            os.environ["RGFM_CONV"] = mode
            errs[mode] = float(np.abs(net(xg, tg).cpu().numpy().astype(np.float64) - g["svhn_f64"]).max())
        assert _engine.range_fallbacks == fb0  # the fp16 path itself produced errs["hx2"]
        if conv_mode_env is None:
            os.environ.pop("RGFM_CONV", None)
        else:
            os.environ["RGFM_CONV"] = conv_mode_env
        arith = {"what": "max |error| of one SVHN U-Net evaluation (B=4) against the reference run in float64",
                 "split_fp16x2_default": errs["hx2"], "split_bf16x3": errs["bx3"], "exact_fp32_mfma": errs["f32"],
                 "reference_own_fp32": float(g["svhn_ref32_err"])}

    ceilings = None
    if rank == 0 and world == 1 and timers:
        # measured ceilings of THIS device, after the timed region: what a register-only f16 MFMA loop sustains
        # under DVFS (the nominal 2500 TFLOP/s assumes 2.4 GHz) and what a float4 copy moves through HBM
        import ctypes
        L = _lib.lib()
        tf, gb = ctypes.c_double(), ctypes.c_double()
        _lib.check(L.rgfm_ubench_hbm_copy(1 << 30, ctypes.byref(gb)))  # (the copy first: the MFMA loop leaves the chip hot)
        _lib.check(L.rgfm_ubench_mfma_f16(ctypes.byref(tf)))
        ceilings = {"mfma_f16_tflops": tf.value, "hbm_copy_gbs": gb.value}

    if rank == 0:
        assert out[0] is not None and out[0].shape[0] == B and torch.isfinite(out[0]).all()
        value = B * args.steps / elapsed
        line = {
            "metric": "paired images/sec (MNIST32+SVHN, 100 Euler steps)",
            "value": value,
            "unit": "paired images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",  # fp32 tensors and accumulation end to end; how the conv products are formed: "arithmetic"
            "arithmetic": ARITHMETIC.get(conv_mode, conv_mode),
            # optional kernel paths of the default arithmetic as this run had them (environment switches)
            "conv_paths": {"upsample_as_parity_classes": conv_mode == "hx2" and os.environ.get("RGFM_UP_T2", "1") != "0",
                           "p_format_hand_over": conv_mode == "hx2" and os.environ.get("RGFM_HX2D", "3") != "0",
                           "winograd": conv_mode == "hx2" and os.environ.get("RGFM_WINO", "0") == "1"},
            "range_fallbacks": _engine.range_fallbacks,  # calls repeated on the split-bf16 convs by the fp16 range guard (0: none)
            "data": "synthetic",
            "config": {
                "workload": "MNIST32 (1x32x32) + SVHN (3x32x32) pair, mc_feng guidance "
                            f"{args.gamma}, {args.euler_steps} Euler steps, N_mc={args.mc}, "
                            f"batch {args.batch_per_gpu} per GPU (BASELINE configs[2]; configs[3] at 8 GPUs)",
                "global_batch": B,
                "parallelism": f"rows sharded over {world} rank(s); all_gather(MC set) + gather(outputs)",
                "step": "one full sampling call (MC pre-phase + ratio + guided Euler loop + gather)",
            },
        }
        if timers and conv_ms > 0:
            ach = conv_fl / (conv_ms * 1e-3) / 1e12
            if conv_mode in PRODUCTS:
                npr = PRODUCTS[conv_mode]
                peak = PEAK_16BIT_MFMA_TFLOPS / npr
                if conv_mode == "hx2":
                    kern = ("conv_mfma_hx2p_kernel / conv_mfma_hx2_kernel (implicit-GEMM conv; fp32 operands as 2 scaled fp16 "
                            "planes, 3 f16-MFMA products per fp32 product, fp32 accumulate)")
                else:
                    kern = ("conv_mfma_bx3w_kernel (implicit-GEMM conv; fp32 operands as 3 exact bf16 planes, 6 bf16-MFMA "
                            "products per fp32 product, fp32 accumulate)")
                basis = (f"fp32-equivalent: {PEAK_16BIT_MFMA_TFLOPS:.0f} TFLOP/s dense 16-bit MFMA / {npr} products; "
                         "achieved counts algorithmic conv FLOPs (2*MAC) once")
                sustained = ceilings["mfma_f16_tflops"] / npr if ceilings else None
            else:
                peak = PEAK_FP32_MFMA_TFLOPS
                kern = "conv_mfma*_kernel (fp32 MFMA implicit-GEMM conv, all shapes)"
                basis = "v_mfma_f32_32x32x2_f32 dense peak"
                sustained = None
            # HBM bytes per conv launch: bench.py cannot collect PMC itself.  The value comes from the rocprofv3
            # --pmc passes of tools/pmc_traffic.py (FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 corrections),
            # committed under profiles/ with the hash of the kernel sources it was taken with; it is reported only
            # while those sources are unchanged, and the line names the file and its sha256.
            traffic, traffic_src = None, None
            tpath = os.path.join(ROOT, "profiles", f"r04_conv_traffic_{conv_mode}.json")
            if os.path.exists(tpath) and args.batch_per_gpu == 512:
                import hashlib
                with open(tpath, "rb") as f:
                    raw = f.read()
                tj = json.loads(raw)
                if tj.get("kernel_sources_sha256") == kernel_sources_sha():
                    traffic = tj.get("per_launch_avg_bytes")
                    traffic_src = {"file": os.path.relpath(tpath, ROOT), "sha256": hashlib.sha256(raw).hexdigest()}
                else:
                    traffic_src = {"file": os.path.relpath(tpath, ROOT), "stale": "kernel sources changed since the PMC pass"}
            line["roofline"] = {
                "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src,
                "peak_nominal": peak, "peak_sustained": sustained,
                "frac_sustained": (ach / sustained) if sustained else None,
                "kernel": kern, "peak_basis": basis, "conv_arithmetic": conv_mode,
                # `achieved` counts the REFERENCE's conv FLOPs.  On the default arithmetic the three Upsample convs
                # (nearest x 2 + 3x3: 13 % of those FLOPs) run as the equivalent ConvTranspose2d(4, 2, 1), 4 / 9 of the products
                # (RGFM_UP_T2): what the matrix cores execute is this fraction of the algorithmic count
                "executed_over_algorithmic": (0.928 if (conv_mode == "hx2" and os.environ.get("RGFM_UP_T2", "1") != "0") else 1.0),
                "launches": int(conv_n), "avg_launch_us": 1e3 * conv_sum_ms / max(conv_n, 1),
                "busy_ms": conv_ms, "sum_launch_ms": conv_sum_ms,
                "timing": "hipEvents (created before the timed region) on the launch streams; achieved = algorithmic "
                          "FLOPs / UNION of the launches' intervals (the two nets of a step run on two streams, so "
                          "launches overlap; sum_launch_ms double-counts that time; RGFM_OVERLAP=0 serialises)",
                "kernel_time_share": conv_ms * 1e-3 / elapsed,
            }
            lt = os.path.join(ROOT, "profiles", "r04_bench_layers_serial.txt")
            if os.path.exists(lt) and conv_mode == "hx2":
                # (static pointer, not a live number) rocprofv3 per-layer table of one main-loop step with every layer's
                # own roofline max(MFMA ceiling, HBM copy rate): a third of the step is memory-bound, DESIGN.md 4
                line["roofline"]["per_layer_table"] = os.path.relpath(lt, ROOT)
            if ceilings:
                line["roofline"]["measured_ceilings"] = {
                    "mfma_f16_tflops": ceilings["mfma_f16_tflops"], "hbm_copy_gbs": ceilings["hbm_copy_gbs"],
                    "how": "in this process after the timed region: register-only v_mfma_f32_32x32x16_f16 loop on random "
                           "operands, 2 waves per SIMD on every CU, 0.25 s (rgfm_ubench_mfma_f16); float4 copy of 1 GiB "
                           "(rgfm_ubench_hbm_copy, read + written bytes)"}
            hk = {}
            # the guidance block is arithmetic-bound, not HBM-bound (its operands are L2-resident): guid_logp is 3 fp32
            # operations per (row, MC sample, element) on the vector ALU (direct differences: SURVEY 7), guid_apply the
            # [B, N] x [N, D] fp32-MFMA GEMM of the weighted sum -- priced against the fp32 peaks, not against HBM
            bnd = args.batch_per_gpu * args.mc * (1024.0 + 3072.0)
            arith_classes = {"guid_logp": ("valu_f32", 3.0 * bnd, PEAK_FP32_MFMA_TFLOPS),  # (packed fp32 VALU peak = fp32 MFMA peak)
                             "guid_apply": ("mfma_f32", 2.0 * bnd, PEAK_FP32_MFMA_TFLOPS)}
            guided_steps = hbm_classes.get("guid_logp", (0, 0, 0, 0))[2]  # one "guid_logp" scope (distances + weights) per guided step
            for name, (busy, tot, nl, by) in hbm_classes.items():
                if nl > 0 and tot > 0:
                    gbs = by / (tot * 1e-3) / 1e9
                    hk[name] = {"launches": int(nl), "avg_launch_us": 1e3 * tot / nl, "algorithmic_gbs": gbs,
                                "frac_nominal": gbs / PEAK_HBM_GBS,
                                "frac_measured_copy": (gbs / ceilings["hbm_copy_gbs"]) if ceilings else None}
                    if name in arith_classes:
                        bound, fl, peak = arith_classes[name]
                        # per guided STEP: the decoupled loop launches guid_apply once per modality (two scopes a step)
                        per_step_ms = tot / (guided_steps if guided_steps else nl)
                        tf = fl / (1e-3 * per_step_ms) / 1e12
                        hk[name].update({"bound": bound, "per_step_us": 1e3 * per_step_ms, "algorithmic_tflops": tf,
                                         "peak_tflops": peak, "frac_arith": tf / peak})
                    else:
                        hk[name]["bound"] = "hbm"
            if hk:
                line["roofline"]["hbm_kernels"] = {
                    "what": "the kernels of a step that are not the implicit-GEMM convs: algorithmic bytes (SURVEY 8d: every "
                            "tensor once) / sum of launch durations against 8000 GB/s nominal and the float4 copy measured "
                            "above; the guidance block also against its arithmetic bound (\"bound\")",
                    "share_of_call": sum(v[1] for v in hbm_classes.values()) * 1e-3 / elapsed, **hk}
            if alt is not None:
                line["roofline"]["exact_fp32_mode"] = alt
            if arith is not None:
                line["roofline"]["arithmetic_check"] = arith
        if ranks is not None:
            line["ranks"] = ranks
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if parity is not None:
            line["parity_check"] = parity
        print(json.dumps(line), flush=True)
        if parity is not None and not parity["max_abs"] <= PARITY_TOL:
            raise SystemExit(f"parity gate failed: max |difference| {parity['max_abs']:.3e} > {PARITY_TOL:g} on rows "
                             f"{parity['row_indices']} of the timed call")
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
