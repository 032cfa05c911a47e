"""bench.py --gpus N has to be startable exactly like the N = 1 line: without WORLD_SIZE in the environment
it becomes the launcher itself -- one rank per GPU under torch.distributed.run on this node, same command line -- and rank 0
prints the one JSON line.  --dry-launch walks that path without GPUs (gloo), including a real all-reduce over the ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                  # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_spawns_its_own_ranks():
    d = _run(["--gpus", "2", "--dry-launch"])
    assert d["dry_launch"] is True and d["n_gpus"] == 2 and d["launcher"] is True
    assert d["allreduce_check"] == d["expected"] == 3.0          # ranks 0 and 1 both took part: 1 + 2


def test_gpus_1_needs_no_launcher():
    d = _run(["--dry-launch"])
    assert d["n_gpus"] == 1 and d["launcher"] is False


def test_launched_by_the_driver_with_torch_distributed_run():
    """The driver's own form: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N (no second launcher inside)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29563", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    # the keys that make an N > 1 line self-explaining (N = 1 runs 4 streams, N > 1 runs 8 per GPU): bench.multi_gpu_summary
    d = json.loads(lines[0])
    for key in ("single_gpu_same_load", "efficiency_vs_same_load", "allreduce_exposed_us"):
        assert key in d, key
    assert set(d["single_gpu_same_load"]) >= {"value", "ms_per_step", "streams", "steps"}


def test_mismatched_world_size_is_an_error_not_a_silent_single_rank_run():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK")}
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_committed_bench_lines_agree_with_their_rocprof_tables():
    """The judged numbers have two sources that must tell the same story: a bench line's per-kernel device times (HIP events inside the
    run) and the rocprofv3 --kernel-trace --stats table of the same command committed next to it (`profiles/`).  For the configs[4]
    line the dominant kernel's roofline fraction is computed from both (`bench_configs._dominant_kernel`); for the headline line the
    tracer's averages sit a few percent above the events (its own overhead), never below and never far off."""
    import csv, glob, json
    import bench_configs as bc
    prof = os.path.join(ROOT, "profiles")
    lines = sorted(glob.glob(os.path.join(prof, "r[0-9][0-9]_bench_c5.json")))
    assert lines
    kern = json.load(open(lines[-1]))["kernels"]
    fl = 640.0 * (2 * 4 * 1024 * 512 + 2 * 512 * 1024)
    d = bc._dominant_kernel(kern, {"k_fwd_persist_xl": fl, "k_fwd_persist_ms": fl, "k_bwd_persist_xl": fl}, bc.PEAK_BF16_MFMA_TF,
                            "r[0-9][0-9]c5_rocprofv3_kernel_stats.csv", bc.C5_ROCPROF_NAMES)
    assert d["kernel"] in ("k_fwd_persist_xl", "k_fwd_persist_ms", "k_bwd_persist_xl") and d["rocprof"] is not None
    assert 0.9 <= d["frac_rocprof"] / d["frac"] <= 1.02, d
    head = json.load(open(sorted(glob.glob(os.path.join(prof, "r[0-9][0-9]_bench_n1.json")))[-1]))
    table = {r["Name"]: float(r["AverageNs"]) * 1e-3 for r in csv.DictReader(open(sorted(glob.glob(os.path.join(prof, "r[0-9][0-9]_rocprofv3_kernel_stats.csv")))[-1]))}
    for probe, sub in (("k_fwd_persist", "k_fwd_persist<"), ("k_bwd_persist", "k_bwd_persist2"), ("k_fold", "k_fold_bf16x3"), ("k_grads_update", "k_grads(")):
        rp = [v for n, v in table.items() if sub in n]
        assert rp, sub
        ev = head["kernels"][probe]["avg_us"]
        assert 0.97 <= max(rp) / ev <= 1.12, (probe, ev, rp)


def test_pmc_summaries_name_the_kernels_of_the_same_rounds_rocprof_table():
    """Counter passes and the kernel trace of a round come from the same command on the same library (tools/profile.sh): every kernel
    the newest PMC summary of a bench line lists as recurring per minibatch must be in the rocprofv3 table committed under the same
    tag, and the summary says which library it saw (`library_source_hash`; tools/round_check.sh refuses one that is older than csrc/)."""
    import csv, glob, json, re
    prof = os.path.join(ROOT, "profiles")
    newest = {}
    for f in sorted(glob.glob(os.path.join(prof, "r[0-9][0-9]*_pmc_traffic.json"))):
        m = re.match(r"(r\d\d)(.*)_pmc_traffic.json", os.path.basename(f))
        newest[m.group(2)] = (m.group(1), f)                       # suffix ("", "s8", "c5") -> newest round
    assert "" in newest and "s8" in newest and "c5" in newest, newest
    for suffix, (rnd, f) in newest.items():
        d = json.load(open(f))
        if rnd >= "r05":
            assert d.get("library_source_hash"), f
        table = os.path.join(prof, "%s%s_rocprofv3_kernel_stats.csv" % (rnd, suffix))
        assert os.path.exists(table), table
        names = [re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", r["Name"]).replace("klstm::", "")) for r in csv.DictReader(open(table))]
        for kname, v in d["kernels"].items():
            if d.get("minibatches") and v["launches"] >= d["minibatches"] and kname.startswith("k_"):
                assert kname in names, (f, kname)


def test_roofline_accounting_is_surveys_8d():
    """`roofline.achieved` divides ALGORITHMIC work by measured time: the per-frame figures are SURVEY.md 8(d)'s, not what the folded
    kernels execute.  Pinned here as numbers: FLOPs per frame 6 (4C I + 4C R + R C) = 13 056 000 at 40/800/512 (forward 4 352 000 -- what
    one frame of a chain launch advances, either direction), the other layers of configs[3]/[4], 79 072 activation bytes per frame,
    5 P 4 bytes of weights / gradients / momentum per minibatch."""
    import bench as b
    import bench_configs as bc
    assert b.FLOPS_PER_FRAME == 13_056_000 == bc.lstm_flops_per_frame(40, 800, 512)
    assert bc.lstm_flops_per_frame(512, 800, 512) == 22_118_400
    assert bc.lstm_flops_per_frame(40, 1024, 512) == 16_711_680
    assert bc.lstm_flops_per_frame(512, 1024, 512) == 28_311_552
    assert abs(sum(bc.lstm_flops_per_frame(i, 1024, 512) for i in (40, 512, 512)) - 73.3e6) < 0.05e6          # configs[4]'s stack
    for S in (1, 4, 8):
        assert b.kernel_alg_flops("k_fwd_persist", S) == S * 4_352_000 == b.kernel_alg_flops("k_bwd_persist", S)
        assert b.kernel_exec_flops("k_gates_fold", S) > b.kernel_alg_flops("k_gates_fold", S)                 # (the fold executes more than it is credited)
    assert b.ACT_BYTES_PER_FRAME == 79_072
    assert b.WEIGHT_BYTES_PER_MINIBATCH == 5 * bc.n_params(40, 800, 512) * 4 == 43_632_000
    assert b.ACT_BYTES_PER_FRAME * 80 + b.WEIGHT_BYTES_PER_MINIBATCH == 49_957_760                            # the 49.96 MB the traffic ratio divides by
