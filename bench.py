#!/usr/bin/env python
"""bench.py -- frames/sec for forward + BPTT + update of one LstmProjectedStreams layer
(40 -> cell 800 / proj 512), BASELINE.json's metric.

A "step" is one BPTT minibatch: Reset (at utterance starts) -> Propagate -> Backpropagate ->
Update over T=20 frames x S streams (bd-nnet-train-lstm-streams.cc:209-228), on synthetic
1000-frame utterances that are already resident in HBM.
  N = 1 : BASELINE.json configs[1] (NumStream = 4 on one GPU).
  N > 1 : BASELINE.json configs[2]'s partitioning: 8 streams per GPU (64 streams on 8 GPUs), independent utterance
          streams sharded over ranks (weak scaling) with ONE all-reduce of the 8.73 MB gradient blob per minibatch (RCCL).
Timing: after W warm-up steps, exactly K steps are timed between barrier + synchronize fences (`first_k_steps`); because
K = 20 steps are only ~5 ms of GPU time, the same K-step block is then repeated back to back until at least
--min-seconds (default 1 s) of timed region have accumulated, and `value` / `ms_per_step` are taken over that whole
region (`timed`).  Max over ranks.  A second run with ragged utterance lengths (900-1100 frames, new utterances enter
at minibatch boundaries, padded frames masked) exercises the mask / Reset path (`ragged`).

`--config` selects the BASELINE.json configuration: c2 (default) = configs[1], the headline; c3 = configs[2]'s per-GPU shard
(the same layer at 8 streams per GPU, = --streams-per-gpu 8); c1 / c4 / c5 = configs[0] / [3] / [4] (bench_configs.py).

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
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
TARGETS_DELAY = 5                                      # train_lstm_streams.sh:7
LR, MOMENTUM, PARAM_SCALE = 1e-5, 0.9, 0.01          # train_lstm_streams.sh:3-4, nnet.proto:3
N_PARAMS = 4 * C_DIM * I_DIM + 4 * C_DIM * R_DIM + 7 * C_DIM + R_DIM * C_DIM            # 2 181 600
# algorithmic FLOPs per frame = 6 * (4C*I + 4C*R + R*C) (fwd + data-grad + weight-grad of the three products)
FLOPS_PER_FRAME = 6 * (4 * C_DIM * I_DIM + 4 * C_DIM * R_DIM + R_DIM * C_DIM)           # 13 056 000
# algorithmic HBM bytes per frame (activations) and per minibatch (weights resident on-chip within a chunk)
ACT_BYTES_PER_FRAME = 4 * (2 * (7 * C_DIM + R_DIM) + 2 * 4 * C_DIM + 2 * R_DIM + 3 * I_DIM)   # 79 072
WEIGHT_BYTES_PER_MINIBATCH = 5 * N_PARAMS * 4                                            # 43 632 000
PEAK_F32_MFMA_TF = 157.3                              # MI355X_MICROARCH.md: f32-input MFMA
PEAK_HBM_TBS = 8.0                                     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BOUNDARY_US = 1.45                                     # MI355X_MICROARCH.md price list: dependent kernel boundary


def make_inputs(S, seed, device):
    """One 1000-frame utterance per stream, N(0,1) features, laid out as 50 time-major
    minibatches [T*S, I]; out_diff ~ N(0, 1e-2) stands in for the (absent) output layers."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    nchunk = UTT_LEN // T_BPTT
    feats = torch.randn(nchunk, T_BPTT * S, I_DIM, generator=g)
    odiff = 0.1 * torch.randn(nchunk, T_BPTT * S, R_DIM, generator=g)
    return feats.to(device), odiff.to(device)


def make_ragged_schedule(S, seed, device, n_utts_per_stream=3):
    """Utterances of 900..1100 frames through the reference's multi-stream batcher (bd-nnet-train-lstm-streams.cc:146-206):
    a stream takes its next utterance at the first minibatch boundary after the previous one ended, the tail of the last
    minibatch of an utterance is padded with its last frame and masked (mask 0 -> out_diff rows are zero), Reset flags mark
    new utterances.  Everything is staged in HBM up front; returns (feats, odiff, flags, valid_frames_per_minibatch)."""
    import kaldi_lstm_amd as k
    rng = np.random.RandomState(seed)
    utts = []
    for u in range(n_utts_per_stream * S):
        ln = int(rng.randint(900, 1101))
        utts.append((rng.randn(ln, I_DIM).astype(np.float32), np.zeros(ln, np.int32)))
    b = k.MultiStreamBatcher(utts, S, T_BPTT, TARGETS_DELAY)
    feats, odiffs, flags, valid = [], [], [], []
    while True:
        nb = b.next()
        if nb is None:
            break
        feat, _tg, mask, fl = nb
        od = (0.1 * rng.randn(T_BPTT * S, R_DIM)).astype(np.float32) * mask[:, None]
        feats.append(feat); odiffs.append(od); flags.append(np.asarray(fl, np.int32)); valid.append(int(mask.sum()))
    return (torch.from_numpy(np.stack(feats)).to(device), torch.from_numpy(np.stack(odiffs)).to(device), flags, valid)


# ---- per-launch figures of the step kernels ------------------------------------------------------------------------
def kernel_alg_flops(name, S):
    """ALGORITHMIC FLOPs of the frames one launch advances: the reference's own products for S frames of that direction
    (...streams.h:246,:275,:312 forward; :391,:408 backward), NOT what a folded kernel executes (W_rm = W_gifo_r W_r_m makes
    the folded kernels contract over the cell axis: more executed FLOPs, fewer launches)."""
    C, R, I = C_DIM, R_DIM, I_DIM
    return {"k_gates_step": 2.0 * S * 4 * C * (R + I), "k_proj_step": 2.0 * S * R * C,
            "k_dr_step": 2.0 * S * 4 * C * (R + I), "k_dm_step": 2.0 * S * R * C,
            "k_gates_fold": 2.0 * S * (4 * C * (R + I) + R * C), "k_dmf_step": 2.0 * S * (4 * C * R + R * C),
            "k_fwd_persist": 2.0 * S * (4 * C * (R + I) + R * C),      # :246 + :275 + :312, all T frames in the launch
            "k_bwd_persist": 2.0 * S * (4 * C * (R + I) + R * C)}[name]  # :391 + :408 + :457 (d_r, d_m, in_diff)


def kernel_exec_flops(name, S):
    """FLOPs the launch actually executes."""
    C, R, I = C_DIM, R_DIM, I_DIM
    return {"k_gates_fold": 2.0 * S * 4 * C * (C + I), "k_dmf_step": 2.0 * S * C * 4 * C}.get(name, kernel_alg_flops(name, S))


def kernel_alg_bytes(name, S, T):
    """ALGORITHMIC HBM bytes of one launch per SURVEY 8(d): the activation rows the reference's math reads/writes for the S
    frames this launch advances, plus this direction's weights amortised over the T launches of a minibatch (8(d) counts
    each weight tensor once per direction per minibatch: it assumes they stay on-chip within a chunk)."""
    C, R, I = C_DIM, R_DIM, I_DIM
    fwd_act = S * (I + (7 * C + R) + R)                      # x in; G,I,F,O,C,H,M,R out; out rows
    bwd_act = S * ((7 * C + R) + R + 2 * 4 * C + 2 * C + R)  # fwd slab in; out_diff; dgifo(t+1) in + dgifo(t) out; dc in/out; d_r
    w_fwd = (4 * C * (I + R) + 7 * C + R * C) / T
    w_bwd = (4 * C * R + 3 * C + R * C) / T
    return 4.0 * {"k_gates_step": fwd_act + w_fwd, "k_proj_step": S * (C + 2 * R) + R * C / T,
                  "k_dr_step": S * (4 * C + R) + 4 * C * R / T, "k_dm_step": bwd_act + R * C / T,
                  "k_gates_fold": fwd_act + w_fwd, "k_dmf_step": bwd_act + w_bwd,
                  "k_fwd_persist": fwd_act + w_fwd, "k_bwd_persist": bwd_act + w_bwd}[name]


def kernel_streamed_bytes(name, S):
    """Bytes this launch HAS to move in the multi-launch design: its whole weight operand (nothing on-chip survives a kernel
    boundary) plus the activation rows read and written."""
    C, R, I = C_DIM, R_DIM, I_DIM
    w = {"k_gates_step": 4 * C * (R + I), "k_proj_step": R * C, "k_dr_step": (R + I) * 4 * C, "k_dm_step": C * R,
         "k_gates_fold": 4 * C * (C + I), "k_dmf_step": C * 4 * C}.get(name)
    if w is None:
        return None
    act = {"k_gates_step": S * (R + I + C) + 7 * C + S * 7 * C, "k_proj_step": S * C + 2 * S * R,
           "k_dr_step": S * 4 * C + 4 * S * (R + I), "k_dm_step": S * R * 5 + S * C * 10 + 3 * C + S * R + S * C * 5,
           "k_gates_fold": S * (C + I + C) + 7 * C + S * 7 * C, "k_dmf_step": S * 4 * C + S * C * 11 + 3 * C + S * C * 5}[name]
    return 4.0 * (w + act)


ROCPROF_TAG = {"k_gates_step": "k_gates_v", "k_dr_step": "k_dr_v", "k_gates_fold": "k_gates_v", "k_dmf_step": "k_dmf_v",
               "k_proj_step": "k_proj_v", "k_dm_step": "k_dm_v", "k_fwd_persist": "k_fwd_persist", "k_bwd_persist": "k_bwd_persist"}


def rocprof_avg_us(kernel_tag, S):
    """Average duration (us) of the kernels whose name contains `kernel_tag` in the latest committed `rocprofv3 --kernel-trace
    --stats` summary for this stream count (profiles/rNN[sS]_rocprofv3_kernel_stats.csv), or (None, None).  STATIC: not this run."""
    import csv
    pat = "r*s%d_rocprofv3_kernel_stats.csv" % S if S != 4 else "r[0-9][0-9]_rocprofv3_kernel_stats.csv"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
    if not files:
        return None, None
    tot = calls = 0
    for row in csv.DictReader(open(files[-1])):
        if kernel_tag in row["Name"]:
            tot += float(row["TotalDurationNs"]); calls += int(row["Calls"])
    return (tot / calls / 1e3, os.path.relpath(files[-1], ROOT)) if calls else (None, None)


def pmc_profile(S=4):
    """The committed rocprofv3 PMC summary (profiles/rNN_pmc_traffic.json, produced by tools/profile.sh on the GPU box in
    SEPARATE passes -- counters cannot be collected inside a timed run), or None.  STATIC: not measured by this process."""
    docs = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):      # latest round last
        try:
            docs.append((json.load(open(f)), os.path.relpath(f, ROOT)))
        except Exception:
            pass
    docs = [d for d in docs if d[0].get("config") not in ("c4", "c5") and d[0].get("kernels")]      # (those belong to bench_configs.py)
    same = [d for d in docs if d[0].get("streams_per_gpu", 4) == S]
    return same[-1] if same else docs[-1] if docs else (None, None)


def init_params(seed=7):
    """U[-ParamScale, +ParamScale] parameters from a fixed-seed host RNG (InitMatParam/InitVecParam semantics,
    ...streams.h:41-53), GetParams order; identical on every rank."""
    rng = np.random.RandomState(seed)
    return ((rng.rand(N_PARAMS) - 0.5) * 2 * PARAM_SCALE).astype(np.float32)


def cpu_baseline(S, budget_s):
    """The oracle (reference op sequence, un-fused, one GEMM per reference AddMatMat) timed on this host on a bounded sample
    of the same workload.  The reference's CPU path calls cblas_sgemm (kaldi-matrix.cc:160-175): `value` = GEMMs through
    OpenBLAS cblas_sgemm on ONE thread (Kaldi nnet1 is single-threaded outside BLAS), `value_threaded` = the same with the
    BLAS on all physical cores, `value_plain_loops` = the oracle's own triple loops (what the parity tests run).
    The only place bench.py touches oracle/ (bench_configs.py: the cpu_baseline leg of --config c1 likewise)."""
    from oracle.oracle import Oracle, use_openblas
    rng = np.random.RandomState(0)
    x = rng.randn(T_BPTT * S, I_DIM).astype(np.float32)
    od = (0.1 * rng.randn(T_BPTT * S, R_DIM)).astype(np.float32)

    def timed(budget):
        o = Oracle(I_DIM, C_DIM, R_DIM, S, np.float32, threads=1)
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
    phys = set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and cpu_model == "unknown CPU":
                cpu_model = line.split(":", 1)[1].strip()
            if line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            if line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ncores = min(len(phys) or logical, logical)
    nthr = min(ncores, 64)                      # OpenBLAS build limit (MAX_THREADS = 64)
    res = {"unit": "frames/s", "kind": "port", "host_logical_cores": logical, "host_physical_cores": ncores}
    n0, dt0 = timed(max(2.0, budget_s / 4))
    res["value_plain_loops"] = n0 * T_BPTT * S / dt0
    blas = use_openblas(1)
    if blas is None:                            # no OpenBLAS found: the plain loops are all there is
        res.update({"value": res["value_plain_loops"], "cores": 1, "blas": "none (oracle's own loops)"})
        n1, dt1, na, dta = n0, dt0, 0, 0.0
    else:
        n1, dt1 = timed(budget_s)
        res.update({"value": n1 * T_BPTT * S / dt1, "cores": 1, "blas": blas})
        use_openblas(nthr)
        na, dta = timed(max(2.0, budget_s / 3))
        res.update({"value_threaded": na * T_BPTT * S / dta, "cores_threaded": nthr,
                    "threaded_note": "a LOWER bound, not a better baseline: at %d rows per step the reference's products are GEMV-sized "
                                     "(the largest is %d x %d x %d), and OpenBLAS spends more on waking %d threads than they bring -- the "
                                     "1-thread figure is the faster CPU path and is what `value` reports" % (S, S, 4 * C_DIM, R_DIM, nthr)})
        use_openblas(0)
    res["sample"] = (f"{n1} minibatches of {T_BPTT}x{S} frames ({dt1:.1f} s) with cblas_sgemm on 1 thread, {na} ({dta:.1f} s) on {nthr} "
                     f"BLAS threads, {n0} ({dt0:.1f} s) with the oracle's own loops; {cpu_model}, {ncores} physical / {logical} logical "
                     f"cores; oracle/lstmp_oracle.c fp32, un-fused reference op order")
    return res



def multi_gpu_summary(value, world, same_load, allreduce_us, ranks_seen=None):
    """The keys that make an N > 1 line self-explaining (N = 1 runs 4 streams, N > 1 runs 8 per GPU: value(N) / (N * value(1)) is
    NOT an efficiency).  `single_gpu_same_load` = this job's per-GPU load (8 streams) on one GPU with no collective, measured by
    every rank on its own GPU in the same process before the library's communicator exists (slowest rank); `efficiency_vs_same_load`
    = value / (N * that); `allreduce_exposed_us` = time of the one gradient all-reduce per minibatch on the engine's stream (HIP
    events around the collective: nothing overlaps it, the Update waits for it)."""
    return {"single_gpu_same_load": same_load,
            "efficiency_vs_same_load": (value / (world * same_load["value"])) if same_load and same_load.get("value") else None,
            "allreduce_exposed_us": allreduce_us, "ranks_seen": ranks_seen}


def kaldi_adapter_leg(S, steps=400, warmup=50):
    """The as-shipped-in-Kaldi figure: tools/kaldi_adapter_bench (C++: klstm_kaldi::LstmProjectedStreams of include/klstm_component.hpp
    exactly as INTEGRATION.md 2 constructs it -- SetUpdateFollows(true), persist_verify = 1 --, Reset + PropagateFnc + BackpropagateFnc +
    Update per minibatch on pitched device matrices) run as a child process on the same GPU while this process is idle."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "kaldi_adapter_bench")
    try:
        if not os.path.exists(exe):
            import importlib.util
            spec = importlib.util.spec_from_file_location("klstm_build", os.path.join(ROOT, "kaldi-lstm_amd", "build.py"))
            mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
            mod.build_adapter_bench()
        out = {}
        for tag, verify, d2h, spin, dsmall in (("", 1, 0, 1, 1), ("with_d2h_per_minibatch", 1, 1, 1, 1),
                                               ("with_d2h_per_minibatch_hipmemcpy", 1, 1, 1, 0), ("verify_by_stream_sync", 1, 0, 0, 1),
                                               ("persist_verify_0", 0, 0, 1, 1)):
            r = subprocess.run([exe, str(S), str(steps), str(warmup), str(verify), str(d2h), str(spin), str(dsmall)], capture_output=True,
                               text=True, timeout=120)
            if r.returncode != 0:
                raise RuntimeError("rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:]))
            d = json.loads(r.stdout.strip().splitlines()[-1])
            if tag:
                out[tag] = {k_: d[k_] for k_ in ("value", "ms_per_step", "persist_giveups")}
            else:
                out.update(d)
        return out
    except Exception as ex:                               # (the headline does not depend on this leg; the line says what happened)
        return {"error": str(ex)}


def folded_chain(eng):
    """Did the engine's last minibatch run on a chain that needs the fold product W_rm = W_gifo_r W_r_m?"""
    return eng.profile_query("persist_launches")[1] > 0 or eng.S <= 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--streams-per-gpu", type=int, default=0,
                    help="default: 4 on one GPU (BASELINE.json configs[1]), 8 per GPU on N > 1 (configs[2]: 64 streams on 8 GPUs)")
    ap.add_argument("--min-seconds", type=float, default=3.0, help="minimum length of the timed region (whole K-step blocks)")
    ap.add_argument("--config", choices=("c1", "c2", "c3", "c4", "c5"), default="c2",
                    help="BASELINE.json configuration: c2 = configs[1] (headline, default); c3 = configs[2]'s per-GPU shard (8 streams "
                         "per GPU); c1 / c4 / c5 = configs[0] / [3] / [4] on one GPU (bench_configs.py)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the ragged-length run and the 8-streams-per-GPU reference point")
    ap.add_argument("--launch", choices=("auto", "graph", "eager"), default="auto",
                    help="engine option 'graph': hipGraph replay per call (robust against a busy host thread) or plain "
                         "stream launches (no fixed cost per graph launch); auto = time both before the warm-up, keep the faster")
    ap.add_argument("--option", action="append", default=[], help="engine option key=value (A/B experiments), repeatable")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch check only (runs without GPUs): bring the N ranks up exactly as a real run does -- the self-launch under "
                         "torch.distributed.run when WORLD_SIZE is not set -- on the gloo backend, all-reduce one number, print one JSON line")
    ap.add_argument("--collective", choices=("auto", "rccl", "oneshot"), default="auto",
                    help="N > 1: the gradient all-reduce -- ncclAllReduce (klstm_allreduce_grads), the one-shot exchange over peer-mapped "
                         "blobs (klstm_oneshot.hip), or auto = both are checked against each other and timed inside the warm-up, the faster "
                         "one runs the timed steps (`allreduce_ab` in the line)")
    ap.add_argument("--force-collective", action="store_true",
                    help="one rank, but through the N > 1 code path: gradient -> klstm_allreduce_grads on a 1-rank RCCL communicator -> "
                         "momentum -> two-launch Update (what every rank of a multi-GPU run executes, minus the wire)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` started like the N = 1 line: become the launcher -- one rank per GPU under torch.distributed.run
        # on this node, the same command line; rank 0 prints the one JSON line
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d (launch with --nproc-per-node equal to --gpus, or without a launcher)" % (args.gpus, world))
    if args.dry_launch:
        import torch.distributed as dist
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(t)
            dist.barrier()
            total = float(t.item())
            dist.destroy_process_group()
        else:
            total = 1.0
        if rank == 0:
            line = {"dry_launch": True, "n_gpus": world, "launcher": os.environ.get("TORCHELASTIC_RUN_ID") is not None or world > 1,
                    "allreduce_check": total, "expected": world * (world + 1) / 2.0}
            if world > 1:                             # the keys every real N > 1 line carries (values: placeholders of a dry run)
                line.update(multi_gpu_summary(0.0, world, {"value": None, "unit": "frames/s", "ms_per_step": None, "streams": 8, "steps": 0}, None))
            print(json.dumps(line))
        return
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    import kaldi_lstm_amd as k
    if args.config in ("c1", "c4", "c5"):
        if world != 1:
            sys.exit("bench.py --config %s is a single-GPU (per-GPU shard) line" % args.config)
        import bench_configs
        print(json.dumps(bench_configs.RUN[args.config](args, k)))
        return
    S = args.streams_per_gpu or (8 if (world > 1 or args.config == "c3") else 4)
    stream = torch.cuda.Stream()
    ones = np.ones(S, np.int32)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def make_engine(S_):
        e = k.Engine(I_DIM, C_DIM, R_DIM, S_, device=local_rank, stream=stream)
        e.set_params(init_params())   # identical on all ranks
        for kv in args.option:
            key, val = kv.split("=")
            e.set_option(key, int(val))
        return e

    eng = make_engine(S)
    feats, odiff = make_inputs(S, 1234 + rank, "cuda")
    nchunk = feats.shape[0]
    same_load = None
    if world > 1 or args.force_collective:
        # the same per-GPU load on ONE GPU, no collective: every rank on its own GPU, before the library's communicator exists
        # (--force-collective: one rank through exactly this code, so that a one-GPU box exercises it)
        with torch.cuda.stream(stream):
            o_sl = torch.empty(T_BPTT * S, R_DIM, device="cuda"); i_sl = torch.empty(T_BPTT * S, I_DIM, device="cuda")

            def step_sl(i):
                c = i % nchunk
                if c == 0:
                    eng.reset(np.ones(S, np.int32))
                eng.propagate(feats[c], o_sl); eng.backpropagate(feats[c], odiff[c], i_sl, MOMENTUM, 2); eng.update(LR)
            for i in range(20):
                step_sl(i)
            n_sl = 200
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20, 20 + n_sl):
                step_sl(i)
            torch.cuda.synchronize()
            dt_sl = torch.tensor([time.perf_counter() - t0], device="cuda", dtype=torch.float64)
            dist.all_reduce(dt_sl, op=dist.ReduceOp.MAX)
            dt_sl = float(dt_sl.item())
        same_load = {"value": n_sl * T_BPTT * S / dt_sl, "unit": "frames/s", "ms_per_step": dt_sl / n_sl * 1e3, "streams": S, "steps": n_sl,
                     "path": "single-GPU step (fused gradient + momentum + Update, no gradient blob, no collective); slowest of the %d ranks" % world}
        eng.close()
        eng = make_engine(S)                         # (fresh parameters and state for the run proper: identical on all ranks)
    out = torch.empty(T_BPTT * S, R_DIM, device="cuda")
    in_diff = torch.empty(T_BPTT * S, I_DIM, device="cuda")
    # N>1: one all-reduce (sum, fp32) of the 8.73 MB gradient blob per minibatch, issued by libklstm.so itself (RCCL); a run
    # on several GPUs that cannot make the library-owned communicator fails instead of silently measuring something else
    collective_on = world > 1 or args.force_collective
    want_oneshot = collective_on and args.collective in ("auto", "oneshot")
    oneshot_note = None
    # (a rank that cannot set the exchange up -- peer mapping refused, no IPC between these devices -- does not raise on its own:
    #  the ranks agree inside DataParallelLstm, dp.py _setup_oneshot, and everybody runs on RCCL)
    dp = k.DataParallelLstm(eng, force_collective=args.force_collective, require_native=collective_on, oneshot=want_oneshot)
    if want_oneshot and dp.oneshot is None:
        oneshot_note = (dp.oneshot_note or "one-shot exchange could not be set up") + ": RCCL only"
    allreduce_ab = None
    torch.cuda.synchronize()

    def step(i):
        c = i % nchunk                      # new utterances on every stream (lock-step) every 50 chunks
        dp.train_step(feats[c], out, odiff[c], in_diff, MOMENTUM, LR, reset_flags=ones if c == 0 else None)

    def timed_block(fn, i0, n):
        fence()
        t0 = time.perf_counter()
        for i in range(i0, i0 + n):
            fn(i)
        fence()
        return max_over_ranks(time.perf_counter() - t0)

    with torch.cuda.stream(stream):
        # ---- N > 1: which all-reduce (untimed): the same local gradient through ncclAllReduce and through the one-shot exchange,
        # results compared, both timed over 16 steps, the faster one kept -- and RCCL kept, loudly, whenever the one-shot status
        # word is not clean or the sums differ
        if dp.oneshot is not None:
            eng.propagate(feats[0], out); eng.backpropagate(feats[0], odiff[0], in_diff, MOMENTUM, 1)
            blob = eng.grad_blob_tensor()
            local = blob.clone()
            eng.allreduce_grads(dp.comm); eng.synchronize()
            via_rccl = blob.clone()
            blob.copy_(local)
            err, st1 = None, None
            try:
                dp.oneshot.allreduce_engine(eng)
                eng.synchronize()
            except Exception as ex:                  # (a rank that cannot run the exchange says so in `good` below: nobody uses it then)
                err = str(ex)
            st1 = dp.oneshot.status()
            via_one = blob.clone()
            maxrel = float(((via_rccl - via_one).abs().max() / (via_rccl.abs().max() + 1e-30)).item())
            blob.copy_(via_rccl)
            eng.apply_momentum(MOMENTUM); eng.update(LR)
            # (two summation orders of N fp32 numbers: a few 1e-7 of the largest entry at 8 ranks; a rank's contribution missing: ~1/N)
            good = torch.tensor([1 if (err is None and st1 == 0 and maxrel <= 1e-5) else 0], dtype=torch.int32, device="cuda")
            if world > 1:
                dist.all_reduce(good, op=dist.ReduceOp.MIN)
            allreduce_ab = {"max_rel_diff_first_step": maxrel, "oneshot_status": st1, "oneshot_error": err}
            times = {}
            for name in (("rccl", "oneshot") if int(good.item()) else ("rccl",)):
                dp.use_oneshot = name == "oneshot"
                trouble = []

                def guarded(i):
                    # the exchange under test: a rank on which a step raises (its engine reports a timed-out exchange) stops stepping
                    # but keeps the ranks' collectives aligned (the barriers and the max-over-ranks of timed_block); the others' waits
                    # are bounded, so they get there too
                    if trouble:
                        return
                    try:
                        step(i)
                    except Exception as ex:
                        trouble.append(str(ex))
                fn = guarded if name == "oneshot" else step
                for i in range(4):
                    fn(i)
                times[name] = timed_block(fn, 4, 16) / 16 * 1e3
                if name == "oneshot":
                    bad = 1 if trouble else 0
                    try:
                        bad = max(bad, 1 if dp.oneshot.status() != 0 else 0)
                        eng.synchronize()
                    except Exception as ex:
                        trouble.append(str(ex)); bad = 1
                    st2 = torch.tensor([bad], dtype=torch.int32, device="cuda")
                    if world > 1:
                        dist.all_reduce(st2, op=dist.ReduceOp.MAX)
                    if int(st2.item()) != 0:
                        allreduce_ab["oneshot_status"] = "failed inside the warm-up steps" + (": " + trouble[0] if trouble else " (on another rank)")
                        times.pop("oneshot")
            allreduce_ab.update({k_ + "_ms_per_step": v for k_, v in times.items()})
            pick = args.collective if args.collective != "auto" else min(times, key=times.get)
            if pick not in times:
                print("bench.py: the one-shot exchange did not pass its check (%s): the timed steps run on RCCL" % allreduce_ab, file=sys.stderr)
                pick = "rccl"
            dp.use_oneshot = pick == "oneshot"
            allreduce_ab["chosen"] = pick
        elif oneshot_note:
            allreduce_ab = {"chosen": "rccl", "note": oneshot_note}
        # ---- launch mode A/B (untimed, independent of --warmup): 4 + 16 steps per mode
        launch = args.launch
        ab = None
        if launch == "auto":
            tm = []
            for mode in (2, 0):                 # 2: a hipGraph per call, also for one-launch calls (option "graph", klstm.h)
                eng.set_option("graph", mode)
                for i in range(4):
                    step(i)
                tm.append(timed_block(step, 4, 16))
            launch = "graph" if tm[0] <= tm[1] else "eager"
            ab = {"graph_ms_per_step": tm[0] / 16 * 1e3, "eager_ms_per_step": tm[1] / 16 * 1e3}
        eng.set_option("graph", 2 if launch == "graph" else 0)
        # ---- warm-up, then exactly K steps, then whole K-step blocks until --min-seconds of timed region
        K = args.steps

        region_trouble = []

        def region_step(i):
            # with the (experimental) one-shot exchange in the timed region: a rank whose engine reports a timed-out exchange stops
            # stepping but keeps the ranks' collectives aligned; the region is void then and is measured again on RCCL (below)
            if region_trouble:
                return
            try:
                step(i)
            except Exception as ex:
                region_trouble.append(str(ex))
        run_step = region_step if (dp.oneshot is not None and dp.use_oneshot) else step
        for i in range(args.warmup):
            run_step(i)

        def timed_region(fn):
            dt_first = timed_block(fn, args.warmup, K)
            blocks = max(0, int(np.ceil((args.min_seconds - dt_first) / max(dt_first, 1e-9))))
            dt_total, nsteps_total = dt_first, K
            if blocks:
                dt_total += timed_block(fn, args.warmup + K, blocks * K)
                nsteps_total += blocks * K
            return dt_first, dt_total, nsteps_total
        dt_first, dt_total, nsteps_total = timed_region(run_step)
        if dp.oneshot is not None and dp.use_oneshot:
            # the one-shot exchange must have come through the whole timed region clean on EVERY rank; otherwise the region is void
            # (a rank that timed out reduced nothing) and is measured again on RCCL
            bad3 = 1 if region_trouble else 0
            try:
                bad3 = max(bad3, 1 if dp.oneshot.status() != 0 else 0)
                eng.synchronize()
            except Exception as ex:
                region_trouble.append(str(ex)); bad3 = 1
            st3 = torch.tensor([bad3], dtype=torch.int32, device="cuda")
            if world > 1:
                dist.all_reduce(st3, op=dist.ReduceOp.MAX)
            allreduce_ab["oneshot_status_after_timed_region"] = int(st3.item())
            if int(st3.item()) != 0:
                print("bench.py: the one-shot exchange failed inside the timed region (%s): measured again on RCCL" %
                      (region_trouble[0] if region_trouble else "on another rank"), file=sys.stderr)
                dp.use_oneshot = False
                allreduce_ab["chosen"] = "rccl (the one-shot exchange failed inside the timed region)"
                for i in range(args.warmup):
                    step(i)
                dt_first, dt_total, nsteps_total = timed_region(step)

        # ---- ragged utterance lengths (900..1100): mask / Reset on the timed path
        ragged = None
        if not args.no_extras:
            rf, rod, rflags, rvalid = make_ragged_schedule(S, 4321 + rank, "cuda")
            nrag = rf.shape[0]

            def rstep(i):
                c = i % nrag
                fl = rflags[c] if c else ones           # wrap-around: every stream starts over
                dp.train_step(rf[c], out, rod[c], in_diff, MOMENTUM, LR, reset_flags=fl if fl.any() else None)
            for i in range(8):
                rstep(i)
            reps = max(1, int(np.ceil(min(args.min_seconds, 0.5) / max(dt_first / K * nrag, 1e-9))))
            dtr = timed_block(rstep, 0, reps * nrag)
            ragged = {"value": reps * sum(rvalid) * world / dtr, "unit": "valid frames/s", "ms_per_step": dtr / (reps * nrag) * 1e3,
                      "minibatches": reps * nrag, "valid_frame_fraction": sum(rvalid) / (nrag * T_BPTT * S),
                      "utterance_frames": "uniform 900..1100", "resets_per_pass": int(sum(int(f.sum()) for f in rflags))}
            del rf, rod

        # ---- roofline leg: per-kernel device time from HIP start/stop events on the engine stream
        NPROF = 10
        base = args.warmup + nsteps_total
        eng.set_option("profile", 1)
        for i in range(3):                   # untimed pass: creates the event pool
            step(base + i)
        eng.profile_query("k_gates_step")
        eng.set_option("profile", 1)         # clears the accumulators, keeps the pool
        for i in range(NPROF):
            step(base + 3 + i)
        kern = {}
        for name in ("k_gates_step", "k_proj_step", "k_dr_step", "k_dm_step", "k_dr_step0", "k_gemm_xproj",
                     "k_gates_fold", "k_gemm_rbatch", "k_reduce_rbatch", "k_gemm_P", "k_reduce_P", "k_dmf_step",
                     "k_gemm_tail", "k_reduce_tail", "k_fold", "k_pack_foldx", "k_fwd_persist", "k_bwd_persist", "k_tail_reduce",
                     "k_grads", "k_grads_update", "k_update_repack", "k_pack", "k_pack_fwd", "k_pack_bwd", "k_apply_momentum",
                     "rccl_allreduce", "oneshot_allreduce"):
            tot, n = eng.profile_query(name)
            if n:
                kern[name] = {"avg_us": tot / n, "launches_per_step": n / NPROF, "us_per_step": tot / NPROF}
        eng.set_option("profile", 0)

        # ---- 8 streams per GPU on this one GPU: the denominator of a 1 -> 8 GPU efficiency (configs[2] runs 8 per GPU)
        s8 = None
        if world == 1 and not args.no_extras and S != 8:
            e8 = make_engine(8)
            e8.set_option("graph", 2 if launch == "graph" else 0)
            f8, o8 = make_inputs(8, 99, "cuda")
            out8 = torch.empty(T_BPTT * 8, R_DIM, device="cuda"); id8 = torch.empty(T_BPTT * 8, I_DIM, device="cuda")
            ones8 = np.ones(8, np.int32)

            def step8(i):
                c = i % nchunk
                if c == 0:
                    e8.reset(ones8)
                e8.propagate(f8[c], out8); e8.backpropagate(f8[c], o8[c], id8, MOMENTUM, 2); e8.update(LR)
            for i in range(10):
                step8(i)
            n8 = 200
            dt8 = timed_block(step8, 10, n8)
            s8 = {"value": n8 * T_BPTT * 8 / dt8, "unit": "frames/s", "ms_per_step": dt8 / n8 * 1e3, "streams": 8, "steps": n8}
            e8.close()

        # ---- what arithmetic produced the number, and the same workload with every product on fp32-range arithmetic
        fold_names = {0: "f32 (v_mfma_f32_16x16x4_f32)", 1: "bf16x3 (three bf16 planes per operand, six products, fp32 accumulate: fp32 range and accuracy)",
                      2: "fp16x2 (two fp16 planes per operand, three products, fp32 accumulate: 22-bit operands; range guard -> bf16x3)"}
        arithmetic = {"recurrence_products": "f32 MFMA (v_mfma_f32_4x4x1_16b_f32 / 16x16x4_f32), fp32 accumulate",
                      "gradient_products": "f32 MFMA", "elementwise": "f32",
                      "fold": fold_names.get(eng.profile_query("fold_mode")[1], "?") if folded_chain(eng) else "none (no fold product on this chain)",
                      "fp16_range_guard_events": eng.profile_query("fp16_redo")[1]}
        # every minibatch of the run was applied: no persistent launch gave up, no Update was left out (the counters of klstm.h "persist")
        tail_wgs = eng.profile_query("persist_tail_wgs")[1]
        ev_names = ("persist_giveups", "persist_replayed", "persist_dropped", "dp_updates_left_out")
        ev = torch.tensor([eng.profile_query(name)[1] for name in ev_names], dtype=torch.int64, device="cuda")
        if world > 1:
            dist.all_reduce(ev, op=dist.ReduceOp.MAX)          # (the largest count on any rank)
        persist_events = dict(zip(ev_names, ev.tolist()))
        strict = None
        if world == 1 and not args.no_extras:
            strict = {}
            for tag, mode in (("fold_f32", 0), ("fold_bf16x3", 1)):
                es = make_engine(S)
                es.set_option("fold_bf16x3", mode)
                es.set_option("graph", 2 if launch == "graph" else 0)

                def step_s(i, es=es):
                    c = i % nchunk
                    if c == 0:
                        es.reset(ones)
                    es.propagate(feats[c], out); es.backpropagate(feats[c], odiff[c], in_diff, MOMENTUM, 2); es.update(LR)
                for i in range(10):
                    step_s(i)
                ns = 400
                dts = timed_block(step_s, 10, ns)
                strict[tag] = {"value": ns * T_BPTT * S / dts, "unit": "frames/s", "ms_per_step": dts / ns * 1e3, "steps": ns,
                               "fold": fold_names[mode]}
                es.close()

    # ---- the other BASELINE.json configurations and the Kaldi-side adapter as secondary legs of the driver-timed line
    legs, adapter = {}, None
    if world == 1 and not args.no_extras and args.config == "c2" and S == 4:
        import bench_configs
        for tag, st_ in (("c4", 200), ("c5", 1000)):          # (run_c5 times steps // 5 minibatches)
            la = argparse.Namespace(steps=st_, warmup=20, min_seconds=0.0, option=[], leg=True, no_cpu_baseline=True, cpu_seconds=0.0)
            try:
                legs[tag] = bench_configs.RUN[tag](la, k)
            except Exception as ex:
                legs[tag] = {"error": str(ex)}
        torch.cuda.synchronize()
        adapter = kaldi_adapter_leg(S)

    frames_per_step = T_BPTT * S * world
    value = nsteps_total * frames_per_step / dt_total
    ms_per_step = dt_total / nsteps_total * 1e3
    if rank == 0:
        step_kernels = ("k_gates_step", "k_proj_step", "k_dr_step", "k_dm_step", "k_gates_fold", "k_dmf_step",
                        "k_fwd_persist", "k_bwd_persist")
        for n in step_kernels:
            if n in kern:
                us = kern[n]["avg_us"] * 1e-6
                if n.endswith("_persist"):             # one launch advances all T frames of its direction
                    us = us / T_BPTT
                    kern[n]["us_per_recurrence_step"] = us * 1e6
                kern[n]["alg_tflops"] = kernel_alg_flops(n, S) / us / 1e12
                kern[n]["alg_gbs"] = kernel_alg_bytes(n, S, T_BPTT) / us / 1e9
        folded = "k_dmf_step" in kern or "k_bwd_persist" in kern
        # dominant kernel = the step kernel with the most device time per minibatch
        dom = max((n for n in kern if n in step_kernels), key=lambda n: kern[n]["us_per_step"])
        per_step_us = kern[dom].get("us_per_recurrence_step", kern[dom]["avg_us"])
        a_bytes, a_flops = kernel_alg_bytes(dom, S, T_BPTT), kernel_alg_flops(dom, S)
        gbs = a_bytes / (per_step_us * 1e-6) / 1e9
        tflops = a_flops / (per_step_us * 1e-6) / 1e12
        # which roofline: algorithmic intensity of the whole path (8(d): ~165 FLOP/B with weights resident) is far above the
        # ridge, but a step at S streams is a chain of dependent exchanges; the kernel's own intensity decides the label
        hbm_bound = (a_flops / a_bytes) < PEAK_F32_MFMA_TF / PEAK_HBM_TBS
        pmc, pmc_file = pmc_profile(S)
        traffic = traffic_ratio = mfma_busy = None
        alg_bytes_mb = ACT_BYTES_PER_FRAME * T_BPTT * S + WEIGHT_BYTES_PER_MINIBATCH
        if pmc and pmc.get("streams_per_gpu", 4) == S and pmc.get("chain", "") == ("persistent" if dom.endswith("_persist") else "launches"):
            for name, v in pmc["kernels"].items():
                if name.startswith(ROCPROF_TAG[dom]):
                    traffic = v["hbm_bytes_per_launch"]
                    mfma_busy = v.get("mfma_busy_frac")
            if pmc.get("hbm_bytes_per_minibatch"):
                traffic_ratio = pmc["hbm_bytes_per_minibatch"] / alg_bytes_mb
        rp_us, rp_file = rocprof_avg_us(ROCPROF_TAG[dom], S)
        if rp_us and dom.endswith("_persist"):
            rp_us /= T_BPTT
        frac_now = gbs / (PEAK_HBM_TBS * 1e3) if hbm_bound else tflops / PEAK_F32_MFMA_TF
        roof = {"bound": "hbm" if hbm_bound else "mfma",
                # neither roof is near: the kernel is a chain of dependent in-launch exchanges (sync_floor below)
                "limiter": ("latency (in-launch exchange)" if dom.endswith("_persist") else "latency (dependent launches)")
                           if max(gbs / (PEAK_HBM_TBS * 1e3), tflops / PEAK_F32_MFMA_TF) < 0.1 else ("hbm" if hbm_bound else "mfma"),
                "frac_rocprof": (frac_now * per_step_us / rp_us) if rp_us else None,
                "rocprof": {"avg_us_per_step": rp_us, "file": rp_file} if rp_us else None,
                "kernel": dom,
                "achieved": gbs if hbm_bound else tflops, "peak": PEAK_HBM_TBS * 1e3 if hbm_bound else PEAK_F32_MFMA_TF,
                "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": gbs / (PEAK_HBM_TBS * 1e3) if hbm_bound else tflops / PEAK_F32_MFMA_TF,
                "traffic": traffic,
                "traffic_source": ((pmc_file + " (static: separate rocprofv3 --pmc passes, not this run)") if traffic else
                                   "none: the committed PMC passes (profiles/) are for %s streams per GPU on the %s chain, this run is %d streams on %s"
                                   % (pmc.get("streams_per_gpu") if pmc else "?", pmc.get("chain") if pmc else "?", S,
                                      "the persistent chain" if dom.endswith("_persist") else "launch-per-step kernels")),
                "avg_us": kern[dom]["avg_us"], "us_per_recurrence_step": per_step_us,
                # one launch of a persistent kernel advances all T recurrence steps of its direction
                "alg_bytes_per_recurrence_step": a_bytes, "alg_flops_per_recurrence_step": a_flops,
                "alg_bytes_per_launch": a_bytes * (T_BPTT if dom.endswith("_persist") else 1),
                "alg_flops_per_launch": a_flops * (T_BPTT if dom.endswith("_persist") else 1),
                "mfma_tflops": tflops, "mfma_frac": tflops / PEAK_F32_MFMA_TF, "hbm_gbs": gbs, "hbm_frac": gbs / (PEAK_HBM_TBS * 1e3),
                "mfma_busy_frac_pmc": mfma_busy,
                # SURVEY 8(d) whole-path figures on ms_per_step (weights resident within a chunk)
                "algorithmic": {"flops_per_minibatch": FLOPS_PER_FRAME * T_BPTT * S, "bytes_per_minibatch": alg_bytes_mb,
                                "mfma_frac": FLOPS_PER_FRAME * T_BPTT * S / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TF,
                                "hbm_frac": alg_bytes_mb / (ms_per_step * 1e-3) / 1e12 / PEAK_HBM_TBS},
                "traffic_ratio": traffic_ratio,
                # 8(d): us per recurrence step against the 2-boundary floor of the reference-shaped chain
                "sync_floor": {"us_per_step_and_direction": per_step_us, "two_boundary_floor_us": 2 * BOUNDARY_US}}
        sb = kernel_streamed_bytes(dom, S)
        if sb:                                  # what a launch-per-step design must stream (its whole weight operand per launch)
            roof["streamed"] = {"bytes_per_launch": sb, "gbs": sb / (per_step_us * 1e-6) / 1e9,
                                "frac": sb / (per_step_us * 1e-6) / 1e9 / (PEAK_HBM_TBS * 1e3),
                                "exec_flops_per_launch": kernel_exec_flops(dom, S)}
        chain_us = sum(v["us_per_step"] for n, v in kern.items() if n in step_kernels)
        res = {
            "metric": "frames/sec fwd+BPTT, 40in/800cell/512proj LSTM at 1/2/4/8 MI355X",
            # `steps` = the steps of the timed region (the K asked for, then whole K-blocks until --min-seconds): steps x ms_per_step = timed_region_s
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": nsteps_total, "steps_requested": args.steps, "warmup": args.warmup,
            "timed_region_s": dt_total,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "google/ LstmProjectedStreams 40->cell800/proj512, NumStream=%d per GPU (%d in total), "
                                   "T_bptt=20, 1000-frame synthetic utterances, fwd+BPTT+update (BASELINE.json %s)"
                                   % (S, S * world, "configs[1]" if world == 1 and S == 4 else
                                      "configs[2]: 64 streams = 8 per GPU x 8 GPUs%s" % ("; this line: one GPU's shard" if world == 1 else "")
                                      if S == 8 else "custom stream count"),
                       "streams_per_gpu": S, "total_streams": S * world, "bptt": T_BPTT,
                       "frames_per_step": frames_per_step, "launch": launch, "launch_ab": ab,
                       "arithmetic": arithmetic,
                       "recurrence": ("persistent weights-resident chain" if "k_bwd_persist" in kern or "k_fwd_persist" in kern else
                                      "folded (W_rm = W_gifo_r W_r_m, one kernel per step and direction)" if folded
                                      else "reference-shaped (gates + projection, d_r + d_m kernels per step)"),
                       "bptt_tail": ("d_r / in_diff from %d tail workgroups of the BPTT launch (one per 32-cell slot and column part, off the chain) + k_tail_reduce"
                                     % tail_wgs if tail_wgs else "d_r / in_diff on the chain's workgroups or as batched products"),
                       "update": ("gradient products + momentum + Update as one pass (klstm_backpropagate with KLSTM_BPTT_FUSE_UPDATE: the "
                                  "Update follows immediately, as in Kaldi's Component::Backpropagate)" if "k_grads_update" in kern else
                                  "gradient products, all-reduce, momentum + Update" if (world > 1 or args.force_collective) else "gradient products, then Update"),
                       "parallelism": "dp%d over streams, 1 all-reduce/minibatch" % world if world > 1 else "single GPU",
                       "collective": dp.collective_in_use(), "ranks_seen": dp.ranks_seen},
            "timed": {"steps": nsteps_total, "seconds": dt_total},
            "first_k_steps": {"steps": K, "seconds": dt_first, "ms_per_step": dt_first / K * 1e3,
                              "value": K * frames_per_step / dt_first},
            "roofline": roof,
            "whole_path_tflops": value * FLOPS_PER_FRAME / 1e12 / world,
            "chain_us_per_step": chain_us, "non_chain_us_per_step": sum(v["us_per_step"] for v in kern.values()) - chain_us,
            "kernels": kern,
        }
        for nm in ("rccl_allreduce", "oneshot_allreduce"):   # exposed time of the gradient all-reduce per minibatch (events on the engine's stream)
            if nm in kern:
                res["allreduce_us"] = kern[nm]["us_per_step"]
        if ragged:
            res["ragged"] = ragged
        if s8:
            res["s8_per_gpu"] = s8
        for tag, leg in legs.items():                  # BASELINE.json configs[3] / configs[4], per-GPU shard on this GPU (bench.py --config c4 | c5)
            res[tag] = leg
        if adapter:
            res["kaldi_adapter"] = adapter
        if world > 1 or args.force_collective:
            res.update(multi_gpu_summary(value, world, same_load, res.get("allreduce_us"), dp.ranks_seen))   # (first measurement of SURVEY 8(e): top-level keys)
        if strict:
            # the headline workload with the fold product on fp32 operands (every product of the path then is an fp32 MFMA) and on
            # three bf16 planes (fp32 range, matrix cores): `value` itself runs what config.arithmetic.fold says
            res["strict_f32"] = strict["fold_f32"]
            res["fold_bf16x3"] = strict["fold_bf16x3"]
        if allreduce_ab:
            res["allreduce_ab"] = allreduce_ab
        res["persist_events"] = persist_events
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(S, args.cpu_seconds)
        line = json.dumps(res)
    eng.close()
    if dist is not None:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    # the ONE JSON line is the last thing on stdout: RCCL prints a version banner to stdout when its communicators go away
    sys.stdout.flush()
    if rank == 0:
        print(line)
        sys.stdout.flush()
    if dist is not None:
        sys.stderr.flush()
        os._exit(0)                        # (no library exit handler writes behind the line)


if __name__ == "__main__":
    main()
