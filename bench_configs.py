"""bench.py --config c1|c4|c5: the other BASELINE.json configurations as driver-runnable lines (per-GPU shard on ONE GPU).

  c1  BASELINE.json configs[0]: standard/ LstmProjected 40 -> cell 800 / proj 512, ONE 1000-frame utterance through the
      nnet-forward path (standard/nnet/nnet-lstm-projected.h:222-316).  The reference case is CPU plumbing: `cpu_baseline` is
      the oracle's forward on this host, `value` the same utterance through the engine (S = 1, whole utterance in one call).
  c4  configs[3]: 2 stacked LstmProjectedStreams (40 -> 512 -> 512, cell 800) + AffineTransform 16624 + Softmax +
      Xent::EvalMasked, NumStream 32 over 8 GPUs = 4 streams per GPU (README.md:24-29, nnet.proto:1-6).
  c5  configs[4]: 3 x LstmProjectedStreams cell 1024 / proj 512 (40 -> 512 -> 512 -> 512), NumStream 256 over 8 GPUs = 32
      streams per GPU, bf16 operands with fp32 accumulate and fp32 masters (build extension; the reference is fp32 only).
(c2 = the headline line, c3 = the headline layer at 8 streams per GPU: `bench.py --streams-per-gpu 8`; both live in bench.py.)

Each function returns the dict bench.py prints: metric / value / unit / ms_per_step / dtype / config.workload / roofline
(whole minibatch: algorithmic FLOPs per SURVEY.md 8(d) over the measured step, against the dense MFMA peak of the dtype)
/ sections (device time per part, HIP events) / kernels (engine probes where an engine runs them).
"""
import time

import numpy as np
import torch

T_BPTT, LR, MOMENTUM = 20, 1e-5, 0.9
PEAK_F32_MFMA_TF, PEAK_BF16_MFMA_TF, PEAK_HBM_TBS = 157.3, 2500.0, 8.0


def lstm_flops_per_frame(I, C, R):
    return 6 * (4 * C * I + 4 * C * R + R * C)        # fwd + data-grad + weight-grad of the three products


def n_params(I, C, R):
    return 4 * C * I + 4 * C * R + 7 * C + R * C


def init_params(I, C, R, seed, scale=0.01):
    rng = np.random.RandomState(seed)
    return ((rng.rand(n_params(I, C, R)) - 0.5) * 2 * scale).astype(np.float32)


def _timed(step, warmup, K, min_seconds):
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(warmup + i)
    torch.cuda.synchronize()
    dt_first = time.perf_counter() - t0
    blocks = max(0, int(np.ceil((min_seconds - dt_first) / max(dt_first, 1e-9))))
    dt, n = dt_first, K
    if blocks:
        t0 = time.perf_counter()
        for i in range(blocks * K):
            step(warmup + K + i)
        torch.cuda.synchronize()
        dt += time.perf_counter() - t0
        n += blocks * K
    return dt, n, dt_first


class _WholeStep:
    """One train step per call, either issued call by call from Python (eager) or replayed from ONE hipGraph per input chunk
    (the inputs of chunk c are fixed device buffers, so the graph of chunk c is captured once and replayed every time the chunk
    comes round): the whole minibatch -- every engine call, the output layer, the loss, the statistics accumulation -- is a
    single graph launch and the ~30 Python -> C-ABI calls of a step (8-10 us each) disappear from the timeline, which is what
    a C++ trainer sitting on the same C-ABI would see.  Everything runs on one explicit stream (graphs cannot be captured on
    the null stream).  Measured on c4: no gain over call-by-call launches on an explicit stream (0.569 vs 0.564 ms) -- the host
    keeps ahead; what DID cost 0.25 ms per step was issuing everything on the NULL stream (0.817 ms).  Off by default."""

    def __init__(self, raw, stream, nkeys, graphed, after=None):
        self.raw, self.stream, self.nkeys, self.graphed, self.after = raw, stream, nkeys, graphed, after
        self.graphs, self.calls = {}, 0

    def __call__(self, i):
        c = i % self.nkeys
        with torch.cuda.stream(self.stream):
            if not self.graphed or self.calls < 3:                 # (the first calls allocate: eager)
                self.raw(c)
            else:
                g = self.graphs.get(c)
                if g is None:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self.stream, capture_error_mode="relaxed"):
                        self.raw(c)
                    self.graphs[c] = g
                g.replay()
            self.calls += 1
            if self.after is not None:
                self.after(c)

    def prepare(self):
        """capture every chunk's graph outside the timed region"""
        for i in range(self.nkeys + 3):
            self(i)
        torch.cuda.synchronize()


_META_OPTIONS = ("whole_step_graph", "eager_loss", "fuse_single_rank")     # options of the bench itself, not of the engines


def _apply_options(engines, args):
    """--option key=value reaches every engine of the net (A-B runs: persist=0, graph=1, ...)."""
    for kv in args.option:
        key, val = kv.split("=")
        if key not in _META_OPTIONS:
            for e in engines:
                e.set_option(key, int(val))


def _engine_kernels(engines, step, base, nprof=10):
    names = ("k_gates_step", "k_proj_step", "k_dr_step", "k_dm_step", "k_dr_step0", "k_gemm_xproj", "k_gates_fold", "k_gemm_rbatch",
             "k_reduce_rbatch", "k_gemm_P", "k_reduce_P", "k_dmf_step", "k_gemm_tail", "k_reduce_tail", "k_fold", "k_pack_foldx",
             "k_fwd_persist", "k_bwd_persist", "k_tail_reduce", "k_fwd_persist_ms", "k_fwd_persist_xl", "k_fold_ms", "k_bwd_persist_xl", "k_gemm_dr", "k_reduce_dr", "k_gemm_indiff",
             "k_reduce_indiff", "k_grads", "k_grads_update", "k_update_repack", "k_pack",
             "k_apply_momentum")
    for e in engines:
        e.set_option("profile", 1)
    for i in range(3):
        step(base + i)
    for e in engines:
        e.profile_query("k_grads"); e.set_option("profile", 1)
    for i in range(nprof):
        step(base + 3 + i)
    kern = {}
    for li, e in enumerate(engines):
        for nme in names:
            tot, n = e.profile_query(nme)
            if n:
                kern["layer%d.%s" % (li, nme)] = {"avg_us": tot / n, "launches_per_step": n / nprof, "us_per_step": tot / nprof}
        e.set_option("profile", 0)
    return kern


def _sections(parts, nrep=20):
    """Device time of named callables (HIP events on torch's current stream), each averaged over nrep calls."""
    out = {}
    for name, fn in parts:
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(nrep):
            fn()
        b.record(); torch.cuda.synchronize()
        out[name] = a.elapsed_time(b) * 1e3 / nrep
    return out


def _pmc(config):
    """The committed rocprofv3 PMC summary of `bench.py --config <config>` (profiles/rNN<config>_pmc_traffic.json, tools/profile.sh: separate
    counter passes), or None.  STATIC: not measured by this process."""
    import glob, json, os
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]%s_pmc_traffic.json" % config)))
    for f in reversed(files):
        try:
            d = json.load(open(f))
            d["file"] = "profiles/" + os.path.basename(f)
            return d
        except Exception:
            pass
    return None


def _roof(flops_per_step, ms_per_step, peak_tf, dtype, note, config=None, alg_bytes_per_step=None):
    tf = flops_per_step / (ms_per_step * 1e-3) / 1e12
    r = {"bound": "mfma", "kernel": "whole minibatch (sections / kernels below)", "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s",
         "frac": tf / peak_tf, "traffic": None, "dtype_peak": dtype, "note": note}
    pmc = _pmc(config) if config else None
    if pmc and pmc.get("hbm_bytes_per_minibatch"):
        # HBM bytes per minibatch from the FETCH_SIZE / WRITE_SIZE passes (corrected as MI355X_MICROARCH.md prescribes) against the
        # algorithmic bytes of SURVEY.md 8(d): activations per frame + every weight tensor five times per minibatch
        r["traffic"] = pmc["hbm_bytes_per_minibatch"]
        r["traffic_source"] = pmc["file"] + " (static: separate rocprofv3 --pmc passes, not this run)"
        if alg_bytes_per_step:
            r["alg_bytes_per_minibatch"] = alg_bytes_per_step
            r["traffic_ratio"] = pmc["hbm_bytes_per_minibatch"] / alg_bytes_per_step
        r["traffic_per_kernel"] = {k_: v["hbm_bytes_per_launch"] for k_, v in pmc.get("kernels", {}).items() if v.get("launches", 0) >= pmc.get("minibatches", 1)}
    return r


def lstm_alg_bytes(I, C, R, frames):
    """activation bytes per frame (slab rows written + read, derivative rows, in / out rows) and every weight tensor
    five times per minibatch (forward, backward, gradient read + write, update)."""
    act = 4 * (2 * (7 * C + R) + 2 * 4 * C + 2 * R + 3 * I)
    return act * frames + 5 * 4 * n_params(I, C, R)


def _dominant_kernel(kern, alg_flops_per_launch, peak_tf, csv_glob, csv_names):
    """The kernel with the most device time per minibatch (summed over the layers): its ALGORITHMIC flops per launch over its
    average launch duration -- by this run's HIP events (`frac`) and by the committed rocprofv3 --kernel-trace --stats summary of
    the same command (`frac_rocprof`; static: not this run)."""
    import csv, glob, os
    agg = {}
    for key, v in kern.items():
        a = agg.setdefault(key.split(".", 1)[-1], [0.0, 0.0])
        a[0] += v["us_per_step"]; a[1] += v["launches_per_step"]
    cand = {nme: a for nme, a in agg.items() if nme in alg_flops_per_launch}
    if not cand:
        return None
    name = max(cand, key=lambda nme: cand[nme][0])
    avg_us = cand[name][0] / cand[name][1]
    fl = alg_flops_per_launch[name]
    out = {"kernel": name, "avg_us": avg_us, "us_per_step": cand[name][0], "alg_flops_per_launch": fl,
           "achieved": fl / (avg_us * 1e-6) / 1e12, "unit": "TFLOP/s", "peak": peak_tf, "frac": fl / (avg_us * 1e-6) / 1e12 / peak_tf,
           "limiter": "latency (in-launch exchange inside the XCD's L2)", "frac_rocprof": None, "rocprof": None}
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", csv_glob)))
    if files:
        best = None
        for row in csv.DictReader(open(files[-1])):
            if any(c in row["Name"] for c in csv_names.get(name, (name,))) and (best is None or float(row["TotalDurationNs"]) > best[1]):
                best = (float(row["AverageNs"]) * 1e-3, float(row["TotalDurationNs"]), row["Name"])
        if best:
            out["frac_rocprof"] = fl / (best[0] * 1e-6) / 1e12 / peak_tf
            out["rocprof"] = {"avg_us": best[0], "name": best[2], "file": "profiles/" + os.path.basename(files[-1])}
    return out


# ------------------------------------------------------------------------------------------------------------------------
def run_c1(args, k):
    from oracle.oracle import Oracle, use_openblas   # (cpu_baseline leg only: the checker as the CPU reference path)
    I, C, R, T = 40, 800, 512, 1000
    p = init_params(I, C, R, 7)
    rng = np.random.RandomState(1234)
    x = rng.randn(T, I).astype(np.float32)
    stream = torch.cuda.Stream()                        # (launches on the null stream cost more in the runtime: see run_c4)
    e = k.Engine(I, C, R, 1, stream=stream)
    e.set_params(p)
    xd = torch.from_numpy(x).cuda(); out = torch.empty(T, R, device="cuda")
    torch.cuda.synchronize()
    one = np.ones(1, np.int32)

    def step(i):
        e.reset(one)                                   # standard/ LstmProjected: zero initial state per utterance
        e.propagate(xd, out)
    dt, n, dt_first = _timed(step, max(3, args.warmup // 10), max(5, args.steps // 25), args.min_seconds)
    ms = dt / n * 1e3
    e.set_option("profile", 1)
    for i in range(5):
        step(i)
    kern = {}
    for nme in ("k_fwd_persist", "k_fold", "k_gates_step", "k_gates_fold", "k_gemm_rbatch"):
        tot, cnt = e.profile_query(nme)
        if cnt:
            kern[nme] = {"avg_us": tot / cnt, "launches_per_step": cnt / 5}
    e.close()
    fwd_flops = 2 * (4 * C * I + 4 * C * R + R * C)
    res = {"metric": "frames/sec forward (nnet-forward), standard/ LstmProjected 40in/800cell/512proj, one 1000-frame utterance",
           "value": T / (ms * 1e-3), "unit": "frames/s", "n_gpus": 1, "steps": n, "warmup": max(3, args.warmup // 10), "ms_per_step": ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "standard/ LstmProjected 40->cell800/proj512, 1 utterance of 1000 frames, forward only, whole utterance in "
                                  "one call (BASELINE.json configs[0]; the reference case is CPU plumbing: cpu_baseline is the figure it names)",
                      "streams_per_gpu": 1, "frames_per_step": T},
           "roofline": _roof(fwd_flops * T, ms, PEAK_F32_MFMA_TF, "f32", "one stream: 999 dependent in-launch exchanges, latency bound"),
           "kernels": kern}
    if not args.no_cpu_baseline:
        blas = use_openblas(1)
        o = Oracle(I, C, R, 1, np.float32, threads=1); o.set_params(p)
        o.reset([1]); o.propagate(x)
        cnt, t0 = 0, time.perf_counter()
        while True:
            o.reset([1]); o.propagate(x); cnt += 1
            dtc = time.perf_counter() - t0
            if dtc >= args.cpu_seconds and cnt >= 2:
                break
        use_openblas(0)
        res["cpu_baseline"] = {"value": cnt * T / dtc, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d utterances of 1000 frames (%.1f s), oracle forward, GEMMs through %s" % (cnt, dtc, blas or "its own loops")}
    return res


# ------------------------------------------------------------------------------------------------------------------------
def run_c4(args, k):
    I, C, R, NPDF, S, T = 40, 800, 512, 16624, 4, T_BPTT
    rng = np.random.RandomState(21)
    dims_in = [I, R]
    graphed = "whole_step_graph=1" in args.option
    stream = torch.cuda.Stream()
    engines = []
    for l in range(2):
        e = k.Engine(dims_in[l], C, R, S, stream=stream)
        e.set_params(init_params(dims_in[l], C, R, 30 + l))
        engines.append(e)
    _apply_options(engines, args)
    W = torch.from_numpy(((rng.rand(NPDF, R) - 0.5) * 0.02).astype(np.float32)).cuda()
    b = torch.zeros(NPDF, device="cuda")
    layers = [k.LstmDP(e) for e in engines] + [k.AffineDP(W, b, k, stream=stream)]
    # loss statistics accumulate on the device and are read back once per 50-minibatch utterance round (the reference's
    # trainer reports every few thousand frames); --option eager_loss=1: read back every minibatch instead
    lazy = "eager_loss=1" not in args.option
    fused = "fuse_single_rank=0" not in args.option
    net = k.DataParallelNnet(layers, k.SoftmaxXentDP(k, lazy=lazy, stream=stream, accumulate=lazy), alloc=lambda n: torch.zeros(n, device="cuda"),
                             fuse_single_rank=fused)
    nchunk = 50
    feats = torch.randn(nchunk, T * S, I, device="cuda")
    tg = torch.from_numpy(rng.randint(0, NPDF, (nchunk, T * S)).astype(np.int32)).cuda()
    mask = torch.ones(T * S, device="cuda")
    ones = [1] * S

    seen = []

    def raw(c):
        net.train_step(feats[c], tg[c], mask, MOMENTUM, LR, reset_flags=ones if c == 0 else None)   # (lazy: statistics onto net.loss.totals)

    def after(c):
        if lazy and c == nchunk - 1:
            seen.append(tuple(net.loss.totals.tolist()))                                # one read-back per utterance round
    torch.cuda.synchronize()
    step = _WholeStep(raw, stream, nchunk, graphed and lazy, after)
    step.prepare()
    dt, n, dt_first = _timed(step, args.warmup, args.steps, args.min_seconds)
    ms = dt / n * 1e3
    ms_eager = None
    if step.graphed:                                   # the same steps issued call by call from Python, for the record
        step.graphed = False
        dte, ne, _ = _timed(step, 10, args.steps, min(1.0, args.min_seconds))
        ms_eager = dte / ne * 1e3
    graphed_used = graphed and lazy
    if getattr(args, "leg", False):                    # (a secondary leg of the default bench line: the timed steps only)
        for e in engines:
            e.close()
        fl = lstm_flops_per_frame(I, C, R) + lstm_flops_per_frame(R, C, R) + 6 * NPDF * R
        return {"value": T * S / (ms * 1e-3), "unit": "frames/s", "ms_per_step": ms, "steps": n, "dtype": "f32",
                "mfma_frac": fl * T * S / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TF}
    kern = _engine_kernels(engines, step, args.warmup + n)
    shard = None
    if fused:
        # what a rank of the 8-GPU run executes, minus the wire: pure local gradients into ONE 57.6 MB blob (the all-reduce payload),
        # then momentum + Update as passes of their own (the fused per-layer Update above has no blob to reduce)
        net2 = k.DataParallelNnet(layers, net.loss, alloc=lambda n_: torch.zeros(n_, device="cuda"), fuse_single_rank=False)

        def raw2(c):
            net2.train_step(feats[c], tg[c], mask, MOMENTUM, LR, reset_flags=ones if c == 0 else None)
        step2 = _WholeStep(raw2, stream, nchunk, False, after)
        step2.prepare()
        dt2, n2, _ = _timed(step2, 10, max(20, args.steps // 2), min(1.0, args.min_seconds))
        shard = {"value": T * S / (dt2 / n2), "unit": "frames/s", "ms_per_step": dt2 / n2 * 1e3, "steps": n2,
                 "path": "local gradients into one fused 57.6 MB blob (the payload of the one all-reduce per minibatch), then momentum + Update: "
                         "a rank of the 8-GPU run minus the wire"}
    # device time of the output tail, part by part
    x80 = torch.randn(T * S, R, device="cuda"); aff = layers[-1]; loss = net.loss
    torch.cuda.synchronize()
    aff.stream = loss.stream = None                    # (_sections times on torch's current stream)
    net_out = aff.propagate(x80)
    diff, _, _, _ = loss.eval(net_out, tg[0], mask)
    torch.cuda.synchronize()
    sections = _sections([("affine_propagate", lambda: aff.propagate(x80)),
                          ("softmax", lambda: k.softmax(net_out, loss._post)),
                          ("affine_gradient+in_diff", lambda: aff.backpropagate(x80, diff, True)),
                          ("affine_momentum_update", lambda: aff.apply(MOMENTUM, LR))])
    for e in engines:
        e.close()
    fl = lstm_flops_per_frame(I, C, R) + lstm_flops_per_frame(R, C, R) + 6 * NPDF * R
    return {"metric": "frames/sec fwd+BPTT+update, 2x LstmProjectedStreams(cell 800, proj 512) + AffineTransform 16624 + Softmax/Xent, per-GPU shard",
            "value": T * S / (ms * 1e-3), "unit": "frames/s", "n_gpus": 1, "steps": n, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "2 stacked LstmProjectedStreams (40->512->512, cell 800) + AffineTransform 512->16624 + Softmax + Xent::EvalMasked, "
                                   "NumStream=4 (one GPU's streams of BASELINE.json configs[3]: 32 streams on 8 GPUs), T_bptt=20, %s, "
                                   "loss statistics %s" % ("SINGLE-GPU training step: gradient + momentum + Update as one pass per layer, no gradient blob, no "
                                                           "all-reduce (`multi_gpu_shard_path` = the path a rank of the 8-GPU run takes)" if fused else
                                                           "multi-GPU shard path: local gradients into one fused 57.6 MB blob, then momentum + Update",
                                                           "accumulated on the device, read back every 50 minibatches" if lazy else "read back every minibatch"),
                       "streams_per_gpu": S, "frames_per_step": T * S,
                       "launch": "one hipGraph per minibatch (captured per input chunk)" if graphed_used else
                                 "call by call from Python on one explicit stream (--option whole_step_graph=1: one hipGraph per minibatch; "
                                 "measured equal: 0.569 vs 0.564 ms)",
                       "ms_per_step_call_by_call": ms_eager},
            "roofline": _roof(fl * T * S, ms, PEAK_F32_MFMA_TF, "f32", "86.2 MFLOP per frame (SURVEY.md 8(d))", "c4",
                              lstm_alg_bytes(I, C, R, T * S) + lstm_alg_bytes(R, C, R, T * S) + 5 * 4 * NPDF * (R + 1) + 4 * T * S * (2 * R + 3 * NPDF)),
            "multi_gpu_shard_path": shard, "sections_us": sections, "kernels": kern}


# ------------------------------------------------------------------------------------------------------------------------
class _FixedDiffLoss:
    """configs[4] names no output layer: out_diff ~ N(0, 1e-2) stands in for it (as in the headline line)."""

    def __init__(self, diff):
        self.diff = diff

    def eval(self, net_out, targets, mask):
        return self.diff, 0.0, 0, 0


def run_c5(args, k):
    I, C, R, S, T, NL = 40, 1024, 512, 32, T_BPTT, 3
    dims_in = [I, R, R]
    engines = []
    stream = torch.cuda.Stream()
    for l in range(NL):
        e = k.Engine(dims_in[l], C, R, S, stream=stream)
        e.set_params(init_params(dims_in[l], C, R, 40 + l, scale=0.02))
        e.set_option("bf16", 1)
        engines.append(e)
    _apply_options(engines, args)
    od = 0.1 * torch.randn(T * S, R, device="cuda")
    fused = "fuse_single_rank=0" not in args.option
    layers = [k.LstmDP(e) for e in engines]
    net = k.DataParallelNnet(layers, _FixedDiffLoss(od), alloc=lambda n: torch.zeros(n, device="cuda"), fuse_single_rank=fused)
    nchunk = 50
    feats = torch.randn(nchunk, T * S, I, device="cuda")
    ones = [1] * S

    torch.cuda.synchronize()

    def step(i):
        c = i % nchunk
        net.train_step(feats[c], None, None, MOMENTUM, LR, reset_flags=ones if c == 0 else None)
    dt, n, dt_first = _timed(step, args.warmup, max(20, args.steps // 5), args.min_seconds)
    ms = dt / n * 1e3
    if getattr(args, "leg", False):
        for e in engines:
            e.close()
        fl = sum(lstm_flops_per_frame(dims_in[l], C, R) for l in range(NL))
        return {"value": T * S / (ms * 1e-3), "unit": "frames/s", "ms_per_step": ms, "steps": n, "dtype": "bf16",
                "mfma_frac": fl * T * S / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TF}
    kern = _engine_kernels(engines, step, args.warmup + n)
    shard = None
    if fused:                                          # (as in run_c4: the path a rank of the 8-GPU run takes, minus the wire)
        net2 = k.DataParallelNnet(layers, _FixedDiffLoss(od), alloc=lambda n_: torch.zeros(n_, device="cuda"), fuse_single_rank=False)

        def step2(i):
            c = i % nchunk
            net2.train_step(feats[c], None, None, MOMENTUM, LR, reset_flags=ones if c == 0 else None)
        dt2, n2, _ = _timed(step2, 10, max(20, args.steps // 10), min(1.0, args.min_seconds))
        shard = {"value": T * S / (dt2 / n2), "unit": "frames/s", "ms_per_step": dt2 / n2 * 1e3, "steps": n2,
                 "path": "local gradients into one fused 49 MB blob (the payload of the one all-reduce per minibatch), then momentum + Update"}
    for e in engines:
        e.close()
    fl = sum(lstm_flops_per_frame(dims_in[l], C, R) for l in range(NL))
    roof = _roof(fl * T * S, ms, PEAK_BF16_MFMA_TF, "bf16", "73.3 MFLOP per frame (SURVEY.md 8(d)); weights-resident chains, one per XCD and direction (klstm_persist_xl.hip), batched products around them (klstm_gemm16.hip)",
                 "c5", sum(lstm_alg_bytes(dims_in[l], C, R, T * S) for l in range(NL)))
    # the two chain launches: the reference's recurrent products of the T S frames one launch advances -- forward r(t-1) W_gifo_r^T (:275) and
    # m(t) W_r_m^T (:312); backward dgifo(t+1) W_gifo_r (:391) and d_r(t) W_r_m (:408, the part that is not the batched P)
    chain_fl = float(T * S) * (2 * 4 * C * R + 2 * R * C)
    roof["dominant_kernel"] = _dominant_kernel(kern, {"k_fwd_persist_xl": chain_fl, "k_fwd_persist_ms": chain_fl, "k_bwd_persist_xl": chain_fl},
                                               PEAK_BF16_MFMA_TF, "r[0-9][0-9]c5_rocprofv3_kernel_stats.csv", C5_ROCPROF_NAMES)
    return {"metric": "frames/sec fwd+BPTT+update, 3x LstmProjectedStreams cell 1024 / proj 512, bf16, per-GPU shard",
            "value": T * S / (ms * 1e-3), "unit": "frames/s", "n_gpus": 1, "steps": n, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "3 stacked LstmProjectedStreams cell 1024 / proj 512 (40->512->512->512), NumStream=32 per GPU (the per-GPU shard "
                                   "of BASELINE.json configs[4]: 256 streams on 8 GPUs), T_bptt=20, bf16 operands / fp32 accumulate / fp32 masters; "
                                   "%s" % ("SINGLE-GPU training step (per-layer Update without a gradient blob; `multi_gpu_shard_path` = what a rank of "
                                           "the 8-GPU run executes)" if fused else "multi-GPU shard path (fused gradient blob, separate momentum + Update)"),
                       "streams_per_gpu": S, "frames_per_step": T * S},
            "roofline": roof, "multi_gpu_shard_path": shard, "kernels": kern}


# engine probe -> substrings of the rocprofv3 kernel names it covers (round 4's lines called the per-XCD forward launch by the
# name of the launcher it went through, k_fwd_persist_ms)
C5_ROCPROF_NAMES = {"k_fwd_persist_ms": ("k_fwd_persist_xl", "k_fwd_persist_ms"), "k_fwd_persist_xl": ("k_fwd_persist_xl",),
                    "k_bwd_persist_xl": ("k_bwd_persist_xl",)}

RUN = {"c1": run_c1, "c4": run_c4, "c5": run_c5}
