#!/usr/bin/env python3
"""bench.py -- variational-loss iterations/sec of the hp-VPINN hot path on N MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One step = one full training iteration of BASELINE.json config 4 (Poisson-2D, 16x16 elements,
20x20 GLL points and 10x10 test functions per element, MLP [2,20,20,20,1] tanh, var_form 1):
Taylor-mode MLP forward at all 102 400 quadrature points, per-element projection + residual,
adjoint, reverse pass, boundary term (320 points), TF1 Adam update.  Inputs are synthetic of that
shape (the driver's exact right-hand side, seeded boundary points, seeded Xavier init) and are
resident in HBM before the timed region.  N>1 shards the 256 elements over the ranks (strong
scaling: total work fixed) with one RCCL all-reduce of the packed gradient/loss buffer per step.

Rank 0 prints ONE JSON line (see the contract in DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

CFG4 = dict(N_el_x=16, N_el_y=16, N_test_x=10, N_test_y=10, N_quad=20, N_bound=80, N_residual=100)
LAYERS = [2, 20, 20, 20, 1]
PEAK_FP64_TFLOPS = 78.6   # MI355X FP64 vector = matrix peak (datasheet; half the 157.3 TF FP32 rate of MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def gemm_flops_per_row(layers):
    return 2 * sum(layers[l] * layers[l + 1] for l in range(len(layers) - 1))


def cpu_baseline(setup, theta, iters, vectorized=False):
    """The oracle (reference-structured torch-fp64 restatement, or its vectorised variant) timed on this host."""
    import torch
    # tiny-op graphs get slower with many threads (128 threads: 26 s/iter vs 4 s/iter at 8 on the
    # same box class), so the baseline is pinned to 8 threads -- stated in "cores"
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    from oracle.vpinn_oracle import OracleVPINN2D
    s = setup
    o = OracleVPINN2D(s["X_u_train"], s["u_train"], s["X_f_train"], s["f_train"], s["XY_quad_train"],
                      s["WXY_quad_train"], None, s["F_ext_total"], s["grid_x"], s["grid_y"], s["N_testfcn_total"],
                      s["X_u_train"], s["u_train"], LAYERS, init_params=theta)
    o.vectorized = vectorized
    o.adam_step()  # warm-up (allocations, thread pool)
    t0 = time.time()
    for _ in range(iters):
        o.adam_step()
    dt = time.time() - t0
    how = ("vectorised oracle (all elements batched, projection as one einsum; the strong CPU baseline B of BASELINE.md)"
           if vectorized else
           "reference-structured oracle (per-element Python loop, one reduction per test-function pair, autograd double "
           "backward; baseline A of BASELINE.md)")
    return {"value": iters / dt, "unit": "it/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "%d full iterations of the same config-4 workload, %s, host has %d logical cpus"
                      % (iters, how, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--backend", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-residual-roofline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--residual-elems", type=int, default=1 << 18)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # (test hook: HPV_BENCH_ONE_GPU=1 runs every rank on cuda:0 with a gloo group -- the whole multi-rank flow, incl. the
    #  in-library exchange, on a single-GPU box; throughput is then meaningless)
    one_gpu = os.environ.get("HPV_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    red_dev = "cpu" if one_gpu else "cuda"
    if world > 1 or ("RANK" in os.environ and os.environ.get("HPV_FORCE_DIST") == "1"):
        # one process per GPU over RCCL; HPV_FORCE_DIST=1 under a 1-process torchrun exercises the same path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from hp_vpinns_amd.drivers import poisson2d
    from hp_vpinns_amd.init import xavier_init
    s = poisson2d.setup(**CFG4, with_test_grid=False)
    theta = xavier_init(LAYERS, 1234)
    model = poisson2d.build_model(s, LAYERS, var_form=1, init_params=theta, backend=args.backend, device=local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        model.h.sync()
        torch.cuda.synchronize()

    model.prepare(args.warmup, args.steps)   # multi-GPU: communicator warm-up + graph capture outside the timed region
    model._step(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    model._step(args.steps, False)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel device times (hipEvents on the stream the kernels run on), untimed extra pass ----
    nt = min(args.steps, 100)
    model.h.enable_timing(True)
    model._step(nt, False)
    model.h.sync()
    ktime = {name: model.h.kernel_time_ms(i)[0] for i, name in enumerate(("mlp_fwd", "project", "mlp_bwd"))}
    model.h.enable_timing(False)
    loss3 = model.loss()
    # the L2-error half of the metric, on a 101 x 101 grid of the exact solution, at the parameters reached after all the
    # iterations above (a few thousand Adam steps from the Xavier start: far from converged, see profiles/r01_convergence.txt)
    import numpy as np
    gx = np.linspace(-1, 1, 101)
    Xt = np.stack(np.meshgrid(gx, gx), -1).reshape(-1, 2)
    rel_l2 = model.rel_l2_error(Xt, poisson2d.u_ext(Xt[:, 0:1], Xt[:, 1:2]))
    n_its_done = args.warmup + args.steps + nt

    # ---- weak-scaling probe (multi-GPU only, reported beside the headline number): every rank keeps a full
    #      config-4 shard (256 elements), i.e. the job solves a 16 x 16N-element problem; shows what the exchange costs
    #      when the per-GPU work is not shrunk to the latency floor ----
    weak = None
    if dist is not None and (world > 1 or os.environ.get("HPV_WEAK_PROBE") == "1"):
        sw = poisson2d.setup(**dict(CFG4, N_el_y=CFG4["N_el_y"] * world), with_test_grid=False, assemble="device", device=local_rank)
        mw = poisson2d.build_model(sw, LAYERS, var_form=1, init_params=theta, backend=args.backend, device=local_rank)
        kw = min(args.steps, 1000)
        mw.prepare(args.warmup, kw)
        mw._step(args.warmup, False)
        dist.barrier(); torch.cuda.synchronize()
        tw = time.perf_counter()
        mw._step(kw, False)
        dist.barrier(); torch.cuda.synchronize()
        tw = torch.tensor([time.perf_counter() - tw], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        weak = {"elements": 256 * world, "elements_per_gpu": 256, "steps": kw, "it_per_s": kw / float(tw.item()),
                "element_iterations_per_s": 256 * world * kw / float(tw.item()),
                "note": "not the headline metric: same kernels, problem grown with N (16 x 16N elements)"}
        del mw

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic.json,
    # produced by scripts/gpu_round.sh on this hardware; FETCH_SIZE corrected x2 as the microarch guide says)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            traffic_tab = json.load(f)
    except Exception:
        traffic_tab = {}
    N_local = (model.Nelementx * model.Nelementy * 400) // world  # points per rank
    C, G = 3, gemm_flops_per_row(LAYERS)
    flops = {"mlp_fwd": C * G * N_local, "mlp_bwd": 2 * C * G * N_local}
    proj_fused = ktime["project"] == 0.0
    if proj_fused:
        # element-block mode: the per-element projection (+ adjoint) runs at the head of the reverse kernel; its
        # sum-factorised flops (2 terms x 2 x (20*10*20 + 10*10*20) x 2 = 48 kflop per element) belong to that launch
        flops["mlp_bwd"] += 48000 * (N_local // 400)
    # Algorithmic HBM bytes per launch: the activation store is 140 doubles per point at config 4
    # (s of layer 1; s, z_x, z_y of layers 2 and 3), written once by the forward and read once by the reverse kernel.
    slots, d_in, C_u, n_res_local = 20 * (1 + 3 + 3), 2, 2, (model.Nelementx * model.Nelementy * 100) // world
    n_data_local = 320 if rank == 0 else 0
    npt = N_local + n_data_local
    abytes = {"mlp_fwd": 8 * npt * (slots + d_in + C),                                   # read X, write slots + channels
              "mlp_bwd": 8 * npt * (slots + d_in + C) + (8 * N_local * 2 * C_u + 16 * n_res_local if proj_fused else 0)}
    # (reverse: read slots, X, adjoint channels; fused projection: read C_u channels + F, write C_u adjoint channels + R)
    dom = max(("mlp_fwd", "mlp_bwd"), key=lambda k: ktime[k])

    def roof(k):
        t = ktime[k] * 1e-3
        tf = flops[k] / t / 1e12 if t > 0 else 0.0
        gbs = abytes[k] / t / 1e9 if t > 0 else 0.0
        t_mfma, t_hbm = flops[k] / (PEAK_FP64_TFLOPS * 1e12), abytes[k] / (PEAK_HBM_GBS * 1e9)
        hbm_bound = t_hbm >= t_mfma        # the ceiling that takes longer at peak rate is the one that bounds the kernel
        r = {"kernel": k, "bound": "hbm" if hbm_bound else "mfma",
             "achieved": gbs if hbm_bound else tf, "peak": PEAK_HBM_GBS if hbm_bound else PEAK_FP64_TFLOPS,
             "unit": "GB/s" if hbm_bound else "TFLOP/s",
             "frac": gbs / PEAK_HBM_GBS if hbm_bound else tf / PEAK_FP64_TFLOPS,
             "traffic": traffic_tab.get(k) if world == 1 else None,
             "bytes_per_launch": abytes[k], "flops_per_launch": flops[k], "avg_ms": ktime[k],
             "other_ceiling": {"bound": "mfma" if hbm_bound else "hbm", "achieved": tf if hbm_bound else gbs,
                               "unit": "TFLOP/s" if hbm_bound else "GB/s",
                               "frac": tf / PEAK_FP64_TFLOPS if hbm_bound else gbs / PEAK_HBM_GBS}}
        return r
    out = {
        "metric": "variational-loss iterations/sec", "value": args.steps / dt, "unit": "it/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "Poisson-2D hp-VPINN, 16x16 elements, 20x20 GLL quad/elem, 10x10 test fcns/elem, "
                               "MLP [2,20,20,20,1] tanh, var_form 1, 320 boundary pts, TF1 Adam (BASELINE config 4)",
                   "points": 102400, "residuals": 25600, "params": 921, "backend": model.backend(),
                   "parallelism": "element-sharded dp%d" % world,
                   "exchange": ("in-library p2p mailboxes over xGMI" if getattr(model, "_p2p", False)
                                else ("rccl all-reduce (torch.distributed)" if world > 1 else "none"))},
        "loss_after": float(loss3[0]),
        "rel_l2_error": {"value": rel_l2, "after_iterations": n_its_done, "grid": "101x101",
                         "note": "50 001 iterations reach 4.6e-3 (profiles/r01_convergence.txt)"},
        "kernel_ms": ktime,
        "roofline": dict(roof(dom), projection_fused_into_reverse=proj_fused,
                         note="dominant kernel; algorithmic bytes = activation store (1120 B/point) + coordinates + channels "
                              "(+ the fused projection's channels, F, R); algorithmic flops = 2*C*G*N of the layer products "
                              "(C=3, G=1720/row); the bound is the ceiling with the larger time at peak rate; traffic = PMC HBM "
                              "bytes (profiles/traffic.json); fp64 ubench ceilings on this chip: 47 TFLOP/s v_mfma_f64_16x16x4, "
                              "72 v_mfma_f64_4x4x4_4b, 62 v_fma_f64, not additive"),
        "roofline_other_kernel": roof("mlp_fwd" if dom == "mlp_bwd" else "mlp_bwd"),
    }
    if weak is not None:
        out["weak_scaling_probe"] = weak
    if world == 1 and not args.no_residual_roofline:
        # the per-element projection (residual + adjoint) kernel on a batch larger than the 256 MB
        # Infinity Cache: 2^18 elements of the config-4 element shape, random channels
        ms, by = model.h.bench_projection(args.residual_elems, 10)
        gbs = by / (ms * 1e-3) / 1e9
        out["roofline_residual"] = {"kernel": "project (residual+adjoint)", "bound": "hbm", "achieved": gbs,
                                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                    "traffic": traffic_tab.get("project_scaled"),
                                    "bytes_per_launch": by, "avg_ms": ms, "n_elem": args.residual_elems}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(s, theta, args.cpu_iters)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        out["cpu_baseline_vectorized"] = cpu_baseline(s, theta, 10, vectorized=True)
        out["speedup_vs_cpu_baseline_vectorized"] = out["value"] / out["cpu_baseline_vectorized"]["value"]
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
