#!/usr/bin/env python
"""bench.py -- frames/sec for forward + BPTT + update of one LstmProjectedStreams layer
(40 -> cell 800 / proj 512), BASELINE.json's metric.

A "step" is one BPTT minibatch: Reset (at utterance starts) -> Propagate -> Backpropagate ->
Update over T=20 frames x S streams (bd-nnet-train-lstm-streams.cc:209-228), on synthetic
1000-frame utterances that are already resident in HBM.  N=1 runs BASELINE.json configs[1]
(NumStream=4).  N>1 shards independent streams over ranks (S per GPU fixed -> weak scaling) with
ONE all-reduce of the gradient blob per minibatch (torch.distributed nccl = RCCL).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

I_DIM, C_DIM, R_DIM, T_BPTT, UTT_LEN = 40, 800, 512, 20, 1000
LR, MOMENTUM, PARAM_SCALE = 1e-5, 0.9, 0.01          # train_lstm_streams.sh:3-4, nnet.proto:3
FLOPS_PER_FRAME = 6 * (4 * C_DIM * I_DIM + 4 * C_DIM * R_DIM + R_DIM * C_DIM)   # 13 056 000
PEAK_F32_MFMA_TF = 157.3                              # MI355X_MICROARCH.md: f32-input MFMA


def make_inputs(S, seed, device):
    """One 1000-frame utterance per stream, N(0,1) features, laid out as 50 time-major
    minibatches [T*S, I]; out_diff ~ N(0, 1e-2) stands in for the (absent) output layers."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    nchunk = UTT_LEN // T_BPTT
    feats = torch.randn(nchunk, T_BPTT * S, I_DIM, generator=g)
    odiff = 0.1 * torch.randn(nchunk, T_BPTT * S, R_DIM, generator=g)
    return feats.to(device), odiff.to(device)


PEAK_HBM_TBS = 8.0                                     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def kernel_flops(name, S):
    """FLOPs of one launch (contraction only).  The folded step kernels contract over the cell axis through
    W_rm = W_gifo_r W_r_m: more FLOPs per step than the reference's two products, one launch instead of two."""
    C, R, I = C_DIM, R_DIM, I_DIM
    return {"k_gates_step": 2.0 * S * 4 * C * (R + I), "k_proj_step": 2.0 * S * R * C,
            "k_dr_step": 2.0 * S * 4 * C * (R + I), "k_dm_step": 2.0 * S * R * C,
            "k_gates_fold": 2.0 * S * 4 * C * (C + I), "k_dmf_step": 2.0 * S * C * 4 * C}[name]


def kernel_bytes(name, S):
    """Algorithmic HBM bytes of one launch: the weight operand has to be streamed once per step (nothing
    on-chip survives a kernel boundary) plus the activation rows read and written."""
    C, R, I = C_DIM, R_DIM, I_DIM
    w = {"k_gates_step": 4 * C * (R + I), "k_proj_step": R * C, "k_dr_step": (R + I) * 4 * C, "k_dm_step": C * R,
         "k_gates_fold": 4 * C * (C + I), "k_dmf_step": C * 4 * C}[name]
    act = {"k_gates_step": S * (R + I + C) + 7 * C + S * 7 * C,              # r, x, c(t-1), bias+peepholes | gifo, c, h, m
           "k_proj_step": S * C + 2 * S * R,                                 # m | r, out
           "k_dr_step": S * 4 * C + 4 * S * (R + I),                         # dgifo(t+1) | 4 split-K slabs
           "k_dm_step": S * R * 5 + S * C * 10 + 3 * C + S * R + S * C * 5,  # out_diff + slabs, 10 cell operands | d_r, dgifo, dc
           "k_gates_fold": S * (C + I + C) + 7 * C + S * 7 * C,              # m(t-1), x, c(t-1), bias+peepholes | gifo, c, h, m
           "k_dmf_step": S * 4 * C + S * C * 11 + 3 * C + S * C * 5          # dgifo(t+1), P + 10 cell operands | dgifo, dc
           }[name]
    return 4.0 * (w + act)


def pmc_traffic(kernel_tag):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc_traffic.json), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        kern = json.load(open(files[-1]))["kernels"]
        for name, v in kern.items():
            if name.startswith(kernel_tag):
                return v["hbm_bytes_per_launch"]
    except Exception:
        return None
    return None


def init_params(seed=7):
    """U[-ParamScale, +ParamScale] parameters from a fixed-seed host RNG (InitMatParam/InitVecParam semantics,
    ...streams.h:41-53), GetParams order; identical on every rank."""
    rng = np.random.RandomState(seed)
    n = 4 * C_DIM * I_DIM + 4 * C_DIM * R_DIM + 7 * C_DIM + R_DIM * C_DIM
    return ((rng.rand(n) - 0.5) * 2 * PARAM_SCALE).astype(np.float32)


def cpu_baseline(S, budget_s):
    """The oracle (reference op sequence, un-fused) timed on this host on a bounded sample of the same workload:
    1 thread (Kaldi nnet1 is single-threaded outside BLAS) = `value`, and with the GEMMs threaded over all host cores
    (what a multi-threaded BLAS under Kaldi would give) = `value_threaded`.  The ONLY place bench.py touches oracle/."""
    from oracle.oracle import Oracle
    rng = np.random.RandomState(0)
    x = rng.randn(T_BPTT * S, I_DIM).astype(np.float32)
    od = (0.1 * rng.randn(T_BPTT * S, R_DIM)).astype(np.float32)

    def timed(threads, budget):
        o = Oracle(I_DIM, C_DIM, R_DIM, S, np.float32, threads=threads)
        o.set_params(init_params())
        o.propagate(x); o.backpropagate(x, od, MOMENTUM); o.update(LR)      # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            o.propagate(x); o.backpropagate(x, od, MOMENTUM); o.update(LR)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= budget and n >= 3:
                return n, dt

    cpu_model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nthr = min(ncores, 16)      # the products are GEMV-sized (M = NumStream): more threads only add fork/join cost
    n1, dt1 = timed(1, budget_s)
    na, dta = timed(nthr, max(2.0, budget_s / 3))
    return {"value": n1 * T_BPTT * S / dt1, "unit": "frames/s", "cores": 1, "kind": "port",
            "value_threaded": na * T_BPTT * S / dta, "cores_threaded": nthr, "host_cores": ncores,
            "sample": f"{n1} minibatches of {T_BPTT}x{S} frames ({dt1:.1f} s) on 1 thread, {na} ({dta:.1f} s) on "
                      f"{nthr} OpenMP threads ({cpu_model}, {ncores} logical cores); oracle/lstmp_oracle.c fp32, un-fused reference op order"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--streams-per-gpu", type=int, default=4)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch", choices=("auto", "graph", "eager"), default="auto",
                    help="engine option 'graph': hipGraph replay per call (robust against a busy host thread) or plain "
                         "stream launches (no ~6 us fixed cost per graph); auto = time both during the warm-up, keep the faster")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    import kaldi_lstm_amd as k
    S = args.streams_per_gpu
    stream = torch.cuda.Stream()
    eng = k.Engine(I_DIM, C_DIM, R_DIM, S, device=local_rank, stream=stream)
    eng.set_params(init_params())   # identical on all ranks
    feats, odiff = make_inputs(S, 1234 + rank, "cuda")
    nchunk = feats.shape[0]
    out = torch.empty(T_BPTT * S, R_DIM, device="cuda")
    in_diff = torch.empty(T_BPTT * S, I_DIM, device="cuda")
    ones = np.ones(S, np.int32)
    dp = k.DataParallelLstm(eng)          # N>1: one all-reduce (sum, fp32) of the 8.73 MB gradient blob per minibatch
    torch.cuda.synchronize()

    def step(i):
        c = i % nchunk                      # new utterances on every stream (lock-step) every 50 chunks
        dp.train_step(feats[c], out, odiff[c], in_diff, MOMENTUM, LR, reset_flags=ones if c == 0 else None)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        launch = args.launch
        if launch == "auto" and args.warmup < 20:
            launch = "graph"
        if launch == "auto":                 # inside the untimed warm-up: half the steps per mode, keep the faster
            half, tm = args.warmup // 2, []
            for mode in (1, 0):
                eng.set_option("graph", mode)
                for i in range(4):
                    step(i)
                fence()
                t0 = time.perf_counter()
                for i in range(half):
                    step(i)
                fence()
                tm.append(time.perf_counter() - t0)
            tt = torch.tensor(tm, device="cuda", dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)      # same decision on every rank
            launch = "graph" if float(tt[0]) <= float(tt[1]) else "eager"
        eng.set_option("graph", 1 if launch == "graph" else 0)
        for i in range(args.warmup if args.launch != "auto" or args.warmup < 20 else 8):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(args.warmup, args.warmup + args.steps):
            step(i)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())

        # ---- roofline leg: per-kernel device time from HIP start/stop events on the engine stream
        NPROF = 10
        eng.set_option("profile", 1)
        for i in range(3):                   # untimed pass: creates the event pool
            step(args.warmup + args.steps + i)
        eng.profile_query("k_gates_step")
        eng.set_option("profile", 1)         # clears the accumulators, keeps the pool
        for i in range(NPROF):
            step(args.warmup + args.steps + 3 + i)
        kern = {}
        for name in ("k_gates_step", "k_proj_step", "k_dr_step", "k_dm_step", "k_dr_step0", "k_gemm_xproj",
                     "k_gates_fold", "k_gemm_rbatch", "k_reduce_rbatch", "k_gemm_P", "k_reduce_P", "k_dmf_step",
                     "k_gemm_tail", "k_reduce_tail", "k_fold", "k_pack_foldx",
                     "k_grads", "k_update_repack", "k_pack", "k_pack_fwd", "k_pack_bwd", "k_apply_momentum"):
            tot, n = eng.profile_query(name)
            if n:
                kern[name] = {"avg_us": tot / n, "launches_per_step": n / NPROF, "us_per_step": tot / NPROF}
        eng.set_option("profile", 0)

    frames = args.steps * T_BPTT * S * world
    value = frames / dt
    res = None
    if rank == 0:
        # dominant kernel = the step kernel that carries the most algorithmic FLOPs per minibatch
        # (k_gates_step forward, k_dr_step backward: 2*S*4C*R each); of those, the slower one.
        step_kernels = ("k_gates_step", "k_proj_step", "k_dr_step", "k_dm_step", "k_gates_fold", "k_dmf_step")
        for n in step_kernels:
            if n in kern:
                kern[n]["tflops"] = kernel_flops(n, S) / (kern[n]["avg_us"] * 1e-6) / 1e12
                kern[n]["gbs"] = kernel_bytes(n, S) / (kern[n]["avg_us"] * 1e-6) / 1e9
        folded = "k_dmf_step" in kern
        # dominant kernel = the step kernel with the most device time per minibatch (folded chain: k_gates_fold forward,
        # k_dmf_step backward; reference-shaped chain: k_gates_step / k_dr_step, which carry the FLOPs)
        cand = ("k_gates_fold", "k_dmf_step") if folded else ("k_gates_step", "k_dr_step")
        dom = max((n for n in kern if n in cand), key=lambda n: kern[n]["us_per_step"])
        # roofline side: arithmetic intensity of a step kernel is ~S/2 FLOP/B (weights are re-streamed every step),
        # the ridge is 157.3 TF / 8 TB/s ~ 20 FLOP/B  ->  HBM-bound below S ~ 40, MFMA-bound above
        intensity = kernel_flops(dom, S) / kernel_bytes(dom, S)
        hbm_bound = intensity < PEAK_F32_MFMA_TF / PEAK_HBM_TBS
        tflops = kernel_flops(dom, S) / (kern[dom]["avg_us"] * 1e-6) / 1e12
        gbs = kernel_bytes(dom, S) / (kern[dom]["avg_us"] * 1e-6) / 1e9
        tag = {"k_gates_step": "k_gates_v", "k_dr_step": "k_dr_v", "k_gates_fold": "k_gates_v", "k_dmf_step": "k_dmf_v"}[dom]
        res = {
            "metric": "frames/sec fwd+BPTT, 40in/800cell/512proj LSTM at 1/2/4/8 MI355X",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "google/ LstmProjectedStreams 40->cell800/proj512, NumStream=%d per GPU, "
                                   "T_bptt=20, 1000-frame synthetic utterances, fwd+BPTT+update "
                                   "(BASELINE.json configs[1])" % S,
                       "streams_per_gpu": S, "total_streams": S * world, "bptt": T_BPTT,
                       "frames_per_step": T_BPTT * S * world, "launch": launch,
                       "recurrence": "folded (W_rm = W_gifo_r W_r_m, one kernel per step and direction)" if folded
                                     else "reference-shaped (gates + projection, d_r + d_m kernels per step)",
                       "parallelism": "dp%d over streams, 1 all-reduce/minibatch" % world if world > 1 else "single GPU"},
            "roofline": ({"bound": "hbm", "kernel": dom, "achieved": gbs, "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s",
                          "frac": gbs / (PEAK_HBM_TBS * 1e3)} if hbm_bound else
                         {"bound": "mfma", "kernel": dom, "achieved": tflops, "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s",
                          "frac": tflops / PEAK_F32_MFMA_TF}),
            "whole_path_tflops": value * FLOPS_PER_FRAME / 1e12 / world,
            "kernels": kern,
        }
        res["roofline"].update({"traffic": pmc_traffic(tag) if S == 4 else None, "avg_us": kern[dom]["avg_us"],
                                "bytes_per_launch": kernel_bytes(dom, S), "flops_per_launch": kernel_flops(dom, S),
                                "flop_per_byte": intensity, "mfma_tflops": tflops,
                                "mfma_frac": tflops / PEAK_F32_MFMA_TF})
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(S, args.cpu_seconds)
        print(json.dumps(res))
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
