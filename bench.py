#!/usr/bin/env python3
"""Benchmark of the SGTAPose inference hot path on MI355X (contract: see the task's bench rules).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--height H] [--width W] [--dtype f32|bf16]

``--gpus N`` with N > 1 launches itself: the parent (which never touches the GPU) starts N fresh rank
processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment), rank 0
prints the one JSON line, and the parent exits non-zero if any rank fails.  Under ``python -m
torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` the ranks are the launcher's.

One "step" = one pass of the WHOLE hot path over one batch of synthetic frame pairs per GPU:
two-frame DLA-34 backbone, temporal cross-attention (all six levels), DCN neck, heads with fused
sigmoid, device decode (peaks + gathers + the [B,7,3] key-point tensor), and -- for N > 1 -- the RCCL
all-gather of those key-points.  Inputs are resident in HBM before the timed region.  Workload at N=1:
BASELINE.json configs[1] (batch 32, 2-frame 480x640, fp32); the same per-GPU batch for N>1 (weak scaling:
the batch shards over ranks with no data-path collective except the key-point all-gather).

After the timed region rank 0 (a) re-runs a few LOCAL steps (no collective) with HIP events around every
C-ABI launch, on the launch stream, for the per-kernel durations behind ``roofline``; (b) at N=1 times the
CPU oracle (our restatement of the reference path; the reference's Python cannot travel to the GPU box) on
a bounded sample for ``cpu_baseline`` and checks the benched batch itself against it (``parity``).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import torch  # noqa: E402  (importing torch does not initialise the GPU)

ARGS = ("x", "pre_img", "pre_hm", "repro_hm", "pre_hm_cls", "repro_hm_cls")
PEAK_TFLOPS_F32_MFMA = 157.3        # MI355X_MICROARCH.md: f32-input MFMA = 64 FLOP/clk/SIMD at 2.4 GHz
PEAK_TFLOPS_BF16_MFMA = 2500.0      # dense bf16 MFMA (the guide's ~2.5 PF; AMD's 5 PF figure is 2:1 sparse)
PEAK_HBM_GBS = 8000.0               # HBM3E spec
DTYPE_LABEL = {"f32x3": "f32-emu-f16x3"}      # what the JSON line calls each engine: emulated fp32 never reads as plain f32
PMC_TABLES = {"f32": os.path.join(REPO, "profiles", "r03_pmc_step_by_kernel.csv"),
              "bf16": os.path.join(REPO, "profiles", "r03_pmc_step_by_kernel_bf16.csv"),
              "f16": os.path.join(REPO, "profiles", "r03_pmc_step_by_kernel_f16.csv"),
              "f32x3": os.path.join(REPO, "profiles", "r03_pmc_step_by_kernel_f32x3.csv")}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------------------
# self-launch: one fresh process per GPU, started BEFORE anything in this process touches the GPU
# ---------------------------------------------------------------------------------------------------------
def spawn_ranks(n, timeout_s=1500.0):
    """The parent makes no torch.cuda / HIP call at all (not even a device count): a rank whose LOCAL_RANK has no GPU
    fails in its own torch.cuda.set_device and takes the run down with it."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0 inherits stdout (the JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    t_start = time.perf_counter()
    try:
        pending = list(enumerate(procs))
        while pending:
            if time.perf_counter() - t_start > timeout_s:
                log("ranks still running after %.0f s: terminating them" % timeout_s)
                for _, q in pending:
                    q.terminate()
                rc = rc or 124
                break
            for item in list(pending):
                r, p = item
                code = p.poll()
                if code is None:
                    continue
                pending.remove(item)
                if code != 0:
                    log("rank %d exited with code %d" % (r, code))
                    rc = rc or code
                    for _, q in pending:       # a dead rank would leave the others hanging in a collective
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ---------------------------------------------------------------------------------------------------------
def build_model(dev, seed=0, arch="dlapawdl3new_34", peaked=False, frames=4):
    from util import golden_state_dict
    from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt
    if arch == "dlapawdT_34":                     # T-frame window (configs[4])
        from sgtapose_amd.synthetic import fill_state_dict_
        heads = {"hm": 7, "reg": 2, "wh": 2, "tracking": 2}
        model = create_model(arch, heads, default_head_conv(heads), default_opt(seq_len=frames))
        sd = fill_state_dict_({k: v.clone() for k, v in model.state_dict().items()}, seed)
    elif arch == "dlapawd3_34":                     # three-frame variant
        heads = {"hm": 7, "reg": 2, "wh": 2, "tracking": 2}
        sd = golden_state_dict(seed, keys="model_keys_three.json")
        model = create_model(arch, heads, default_head_conv(heads), default_opt())
    else:
        sd = golden_state_dict(seed, peaked=peaked)
        model = create_model(arch, DEFAULT_HEADS, default_head_conv(), default_opt())
    model.load_state_dict(sd, strict=True)
    return model.to(dev).eval(), sd


def instrumented_pass(run_step, steps):
    """Per-kernel durations with HIP events on the launch stream.  Returns {kernel: dict}."""
    from sgtapose_amd import _lib
    agg = {}
    for _ in range(steps):
        _lib.PROFILE = []
        run_step()
        torch.cuda.synchronize()
        rec, _lib.PROFILE = _lib.PROFILE, None
        for name, meta, e0, e1 in rec:
            key = meta["kernel"] if meta else name
            a = agg.setdefault(key, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0, exec_flops=0.0, peak=None))
            if meta and meta.get("peak_tflops"):
                a["peak"] = meta["peak_tflops"]         # a kernel that runs on another matrix pipe than the engine's dtype
            a["ms"] += e0.elapsed_time(e1)
            a["launches"] += 1
            if meta:
                a["flops"] += meta["flops"]
                a["bytes"] += meta["bytes"]
                a["exec_flops"] += meta.get("exec_flops", meta["flops"])
    for a in agg.values():
        for k in ("ms", "flops", "bytes", "exec_flops"):
            a[k] /= steps
        a["launches"] //= steps
    return agg


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of ``kernel`` from the committed PMC passes (separate ``rocprofv3 --pmc FETCH_SIZE`` /
    ``--pmc WRITE_SIZE`` runs of THIS command, condensed by tools/summarize_step_pmc.py, which also records the
    workload and the commit the passes were made at).  bench.py cannot collect counters itself; the table is used only
    when its workload matches this run's, otherwise traffic is null.  gfx950 correction per the guide: FETCH_SIZE
    tallies the 128-byte requests of 16-B/lane streams at 64 bytes -> doubled."""
    PMC_TABLE = PMC_TABLES.get(workload.get("dtype"), PMC_TABLES["f32"])
    try:
        import csv
        meta = json.load(open(PMC_TABLE + ".meta.json"))
        if any(meta.get(k) != v for k, v in workload.items()):
            return None, "PMC table %s is for %s, this run is %s" % (os.path.basename(PMC_TABLE), meta, workload), None
        for r in csv.DictReader(open(PMC_TABLE)):
            if r["kernel"] == kernel:
                # the family's bytes over ONE step; the caller divides by the launches of its own instrumented pass (the timed
                # step may run a layer as two half-batch launches on two streams, the instrumented pass runs it as one)
                byts = (2.0 * float(r["FETCH_SIZE_MB_raw"]) + float(r["WRITE_SIZE_MB_raw"])) * 1e6
                return byts, ("bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) of the kernel family over one step / launches "
                              "per step, %s (separate --pmc passes of this command; not collected live)"
                              % os.path.relpath(PMC_TABLE, REPO)), meta.get("commit")
    except (OSError, KeyError, ValueError):
        pass
    return None, "no PMC table for this kernel / workload under profiles/", None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


# parity bars per compute dtype: fp32 = north_star's 1e-3 on heat-maps / key-points; the 16-bit paths = their STATED
# tolerances on the logits (tests/test_bf16_parity.py, tests/test_f16_parity.py) -- a 16-bit run is never held to, or
# reported against, the fp32 bar
PARITY_BARS = {"f32": {"max_abs": 1e-3, "text": "1e-3 abs on hm / reg / tracking (north_star)"},
               "f32x3": {"max_abs": 1e-3, "text": "1e-3 abs on hm / reg / tracking (north_star): the fp32 bar, fp32 emulated by operand splitting"},
               "f16": {"max_abs": 1.2e-2, "text": "stated f16 tolerance: 1.2e-2 abs on reg / tracking and on the hm logits"},
               "bf16": {"max_abs": 6e-2, "text": "stated bf16 tolerance: 6e-2 abs on reg / tracking and on the hm logits"}}


def cpu_reference(sd, inp, budget_b8_s=40.0):
    """The CPU oracle (kind 'port') timed on frame pairs OF THE BENCHED BATCH, all host threads: B=1 (median of 5 after
    2 warm-ups, five different samples) and B=8 (median of up to 5 after 1-2 warm-ups, bounded by ``budget_b8_s``).
    -> (cpu_baseline record, {sample index: (oracle output, oracle dets)}): the B=1 outputs are the parity reference."""
    import oracle
    B = inp["x"].shape[0]
    H, W = inp["x"].shape[2:]
    cores = torch.get_num_threads()

    def run(idx):
        a = [inp[k][idx] for k in ARGS]
        t0 = time.perf_counter()
        o = oracle.process_reference(sd, *a)
        return time.perf_counter() - t0, o

    picks = sorted({0, B // 4, B // 2, (3 * B) // 4, B - 1})
    while len(picks) < 5:
        picks.append(picks[-1])
    for _ in range(2):
        run(slice(0, 1))
    t1, refs = [], {}
    for i in picks:
        dt, o = run(slice(i, i + 1))
        t1.append(dt)
        refs[i] = o
    res = {"value": round(1.0 / _median(t1), 4), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
           "cpu": cpu_model(),
           "sample": "B=1: median of %d runs after 2 warm-ups, samples %s of the benched batch (%dx%d) through "
                     "oracle.process_reference (forward + sigmoid + decode), %d torch threads"
                     % (len(t1), picks, H, W, cores)}
    if B >= 8:
        t8, t_start, warm = [], time.perf_counter(), 0
        while len(t8) < 5 and (time.perf_counter() - t_start < budget_b8_s or len(t8) < 2):
            dt, _ = run(slice(0, 8))
            if warm < (2 if dt < 6.0 else 1):
                warm += 1
                continue
            t8.append(dt)
        res["b8"] = {"value": round(8.0 / _median(t8), 4), "unit": "frame-pairs/s", "runs": len(t8), "warmups": warm,
                     "sample": "B=8: samples 0..7 of the benched batch in one oracle call, median of %d runs after %d warm-up(s) "
                               "(bounded by a %.0f s budget)" % (len(t8), warm, budget_b8_s)}
        res["sample"] += "; B=8: %d timed runs after %d warm-up(s)" % (len(t8), warm)
    return res, refs


def batch_parity(refs, hip_out, hip_dets, dtype):
    """The samples in ``refs`` (oracle outputs) against the same samples of the HIP batch (they ran at the full batch size)."""
    import numpy as np
    bar = PARITY_BARS[dtype]
    err = {k: 0.0 for k in ("hm", "hm_logit", "reg", "tracking")}
    found_ref = found_same = int_same = 0
    l2 = 0.0
    hip_kp = hip_dets["kp"]
    for i, (o, d) in refs.items():
        for k in ("hm", "reg", "tracking"):
            err[k] = max(err[k], float((hip_out[k][i:i + 1].cpu() - o[k]).abs().max()))
        lg = lambda t: torch.logit(t.double().clamp(1e-12, 1 - 1e-12))      # noqa: E731
        err["hm_logit"] = max(err["hm_logit"], float((lg(hip_out["hm"][i:i + 1].cpu()) - lg(o["hm"])).abs().max()))
        kp_ref = np.concatenate([d["cts_wreg"][0].reshape(7, 2), d["scores"][0].reshape(7, 1)], 1)
        kp_hip = hip_kp[i].cpu().numpy().astype(np.float64)
        f = kp_ref[:, 2] >= 0
        found_ref += int(f.sum())
        found_same += int(np.array_equal(f, kp_hip[:, 2] >= 0))
        int_same += int(np.array_equal(hip_dets["xs"][i].cpu().numpy(), d["xs"][0]) and
                        np.array_equal(hip_dets["ys"][i].cpu().numpy(), d["ys"][0]))
        if f.any():
            l2 = max(l2, float(np.sqrt(((kp_hip[f, :2] - kp_ref[f, :2]) ** 2).sum(1)).max()))
    worst = max(err["hm"] if dtype in ("f32", "f32x3") else err["hm_logit"], err["reg"], err["tracking"])
    return {"samples_of_benched_batch": sorted(refs), "hm_max_abs": err["hm"], "hm_logit_max_abs": err["hm_logit"],
            "reg_max_abs": err["reg"], "tracking_max_abs": err["tracking"], "keypoints_found_ref": found_ref,
            "found_masks_identical": found_same == len(refs),
            "integer_peaks_identical": int_same == len(refs), "keypoint_l2_max_px": l2,
            "bar": bar["text"] + "; HIP ran these samples inside the full benched batch",
            "within_bar": bool(worst <= bar["max_abs"])}


def peaked_reference(inp, picks):
    """Oracle outputs of ``picks`` with the second weight set (peaked heat-maps)."""
    import oracle
    from util import golden_state_dict
    sd = golden_state_dict(0, peaked=True)
    return {i: oracle.process_reference(sd, *[inp[k][i:i + 1] for k in ARGS]) for i in picks}


def peaked_parity(dev, inp, prefs, dtype):
    """Key-point parity with the second weight set: the benched inputs through a model whose 'hm' head emits isolated
    peaks, HIP at the full batch size vs the oracle outputs ``prefs``.  Per reference-found key-point: same integer peak?
    sub-pixel distance of (x + reg_x, y + reg_y)?"""
    import numpy as np
    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.synthetic import PEAKED_HM_BIAS, PEAKED_HM_GAIN
    model, _ = build_model(dev, peaked=True)
    model.engine_dtype = dtype
    det = BatchedDetector(model)
    out, dets = det.process_device(*[inp[k].to(dev) for k in ARGS])
    kp = det.keypoints(dets).cpu().numpy().astype(np.float64)
    xs, ys, sc = dets["xs"].cpu().numpy(), dets["ys"].cpu().numpy(), dets["scores"].cpu().numpy()
    found = same_kp = same_maps = n = 0
    l2 = hm = 0.0
    for i, (o, d) in prefs.items():
        f = d["scores"][0] >= 0
        found += int(f.sum())
        n += 7
        hm = max(hm, float((out["hm"][i:i + 1].cpu() - o["hm"]).abs().max()))
        same_maps += int(np.array_equal(xs[i], d["xs"][0]) and np.array_equal(ys[i], d["ys"][0]))
        hit = f & (sc[i] >= 0) & (xs[i] == d["xs"][0]) & (ys[i] == d["ys"][0])
        same_kp += int(hit.sum())
        if hit.any():
            l2 = max(l2, float(np.sqrt(((kp[i][hit, :2] - d["cts_wreg"][0].reshape(7, 2)[hit]) ** 2).sum(1)).max()))
    del model, det
    torch.cuda.empty_cache()
    return {"weights": "peaked set (hm.2.weight x %g, bias %g)" % (PEAKED_HM_GAIN, PEAKED_HM_BIAS),
            "samples_of_benched_batch": sorted(prefs), "keypoints_found_ref": found, "keypoints_total": n,
            "keypoints_same_integer_peak": same_kp, "integer_peaks_identical": same_maps == len(prefs),
            "keypoint_l2_max_px_on_shared_peaks": l2, "hm_max_abs": hm}


def dry_run(args):
    """The multi-rank control flow of main() without the GPU: gloo group, one all-gather of fake [B,7,3] key-points per
    step, barrier + max-over-ranks timing, rank 0 prints the JSON line (marked dry_run)."""
    import torch.distributed as dist
    from sgtapose_amd.distributed import gather_keypoints
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = args.batch
    kp = torch.full((B, 7, 3), float(rank))
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        allkp = gather_keypoints(kp, world * B)
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ok = tuple(allkp.shape) == (world * B, 7, 3) and all(float(allkp[r * B, 0, 0]) == r for r in range(world))
    # the checks of the GPU run, outside the timed region: every rank is there and holds the SAME gathered tensor
    cs = allkp.double().sum().reshape(1)
    lo, hi = cs.clone(), cs.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = bool(lo.item() == hi.item())
    ok = ok and same and dist.get_world_size() == args.gpus
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": dist.get_world_size(), "gather_ok": ok,
                          "gathered_keypoints_equal_on_all_ranks": same, "dtype": args.dtype, "arch": args.arch,
                          "steps": args.steps, "warmup": args.warmup}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


# ---------------------------------------------------------------------------------------------------------
def roofline_record(local_step, dtype, wl, el_ms, breakdown=False):
    """Per-kernel durations of a LOCAL (collective-free) instrumented pass -> (roofline of the dominant kernel, per-kernel
    list, kernels without a FLOP / byte model)."""
    # fp16 and bf16 MFMA run at the same rate; the f32x3 engine's remaining fp32 kernels carry their own peak in their meta
    peak = PEAK_TFLOPS_F32_MFMA if dtype in ("f32", "f32x3") else PEAK_TFLOPS_BF16_MFMA
    agg = instrumented_pass(local_step, 2)
    tot = sum(a["ms"] for a in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
    if breakdown:
        log("%-44s %8s %6s %9s %9s" % ("kernel (C-ABI entry / template)", "ms/step", "share", "TFLOP/s", "GB/s"))
        for k, a in rows:
            tf = a["flops"] / a["ms"] / 1e9 if a["flops"] else 0
            gb = a["bytes"] / a["ms"] / 1e6 if a["bytes"] else 0
            log("%-44s %8.3f %5.1f%% %9.1f %9.0f  x%d" % (k, a["ms"], 100 * a["ms"] / tot, tf, gb, a["launches"]))
        log("sum of kernel time %.2f ms/step (event-bracketed), timed region %.2f ms/step" % (tot, el_ms))
    top, ta = next((k, a) for k, a in rows if a["flops"] or a["bytes"])
    peak_engine = peak
    peak = ta["peak"] or peak_engine
    eff = ta["flops"] / ta["ms"] / 1e9                     # TFLOP/s, algorithmic (direct-form, un-padded) flops
    exe = ta["exec_flops"] / ta["ms"] / 1e9                # TFLOP/s actually issued to the matrix pipe
    gbs = ta["bytes"] / ta["ms"] / 1e6                     # GB/s of algorithmic (compulsory) bytes
    mfma_frac, hbm_frac = exe / peak, gbs / PEAK_HBM_GBS
    bound = "hbm" if hbm_frac > mfma_frac else "mfma"
    traffic, tnote, tcommit = pmc_traffic(top, wl)
    if traffic is not None:
        traffic = round(traffic / max(1, ta["launches"]))
    roof = {
        "kernel": top, "bound": bound,
        "achieved": round(gbs if bound == "hbm" else exe, 2), "peak": PEAK_HBM_GBS if bound == "hbm" else peak,
        "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
        "frac": round(hbm_frac if bound == "hbm" else mfma_frac, 4),
        "traffic": traffic, "traffic_note": tnote, "traffic_commit": tcommit,
        "effective_tflops": round(eff, 2), "executed_tflops": round(exe, 2), "mfma_frac": round(mfma_frac, 4),
        "alg_GBs": round(gbs, 1), "hbm_frac": round(hbm_frac, 4),
        "note": "frac = work actually issued / peak of the binding unit (never > 1): executed matrix FLOPs (a Winograd "
                "kernel issues 16/36 or 36/144 of the direct-form multiplies) / MFMA peak, or algorithmic bytes / HBM "
                "peak; effective_tflops = direct-form algorithmic FLOPs / time.  The peaks are nominal-clock peaks: with real "
                "data these kernels run power-limited at ~0.8x the clock they reach on all-zero data "
                "(profiles/r03_conv_data_dependence.txt)",
        "avg_launch_ms": round(ta["ms"] / ta["launches"], 4), "launches_per_step": ta["launches"],
        "share_of_step": round(ta["ms"] / tot, 3)}
    kernels = []
    for k, a in rows:
        if a["flops"]:
            kernels.append({"kernel": k, "ms_per_step": round(a["ms"], 3),
                            "tflops": round(a["flops"] / a["ms"] / 1e9, 2),
                            "executed_tflops": round(a["exec_flops"] / a["ms"] / 1e9, 2),
                            "mfma_frac": round(a["exec_flops"] / a["ms"] / 1e9 / (a["peak"] or peak_engine), 4),
                            "mfma_peak_tflops": a["peak"] or peak_engine,
                            "alg_GBs": round(a["bytes"] / a["ms"] / 1e6, 1),
                            "hbm_frac": round(a["bytes"] / a["ms"] / 1e6 / PEAK_HBM_GBS, 4)})
    other = {k: round(a["ms"], 3) for k, a in rows if not a["flops"]}
    return roof, kernels, other


def timed_steps(run_step, barrier, steps, warmup):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize -> (seconds, host enqueue seconds)."""
    for _ in range(warmup):
        run_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step()
    enq = time.perf_counter() - t0                  # host time to ENQUEUE the K steps (the launches are asynchronous)
    barrier()
    return time.perf_counter() - t0, enq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="frame pairs per GPU per step")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--arch", default="dlapawdl3new_34", choices=["dlapawdl3new_34", "dlapawd3_34", "dlapawdT_34"],
                    help="dlapawd3_34 = the reference's three-frame variant (frame triples/s); dlapawdT_34 = its generalisation "
                         "to a window of --frames frames (BASELINE configs[4]: --frames 4); neither is the headline metric")
    ap.add_argument("--frames", type=int, default=4, help="temporal window length of dlapawdT_34 (previous frames + current)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16", "f32x3"],
                    help="f32 = BASELINE configs[1] (the headline); bf16 = configs[2]'s data path (bf16 feature maps / "
                         "weights on bf16 MFMA, fp32 accumulate / softmax / LayerNorm / decode); f16 = the same data path "
                         "on IEEE half (an eighth of bf16's rounding error, same speed); f32x3 = fp32 ACCURACY with the backbone "
                         "/ head convolutions emulated on the 16-bit matrix pipe by operand splitting (engine_x3.py)")
    ap.add_argument("--secondary", default=None,
                    help="comma list of further compute dtypes timed in the same process after the primary one and reported "
                         "under 'secondary' (default: f32x3,bf16,f16 for the default single-GPU fp32 two-frame workload, else none)")
    ap.add_argument("--sustained-steps", type=int, default=60,
                    help="a second, longer timed region of the primary dtype reported as sustained_value (0: skip)")
    ap.add_argument("--mode", default="batch", choices=["batch", "stream"],
                    help="stream = the reference's operating point: ONE camera, frame after frame, closed loop on the device "
                         "(pre-process -> network -> decode -> post-process -> PnP -> next priors); reports ms/frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--early-levels", default=None,
                    help="comma list of pyramid levels whose attention overlaps the backbone on a side stream "
                         "(engine default: 2,3; empty string: none)")
    ap.add_argument("--skip-dead-levels", action="store_true",
                    help="do not compute temporal levels 0/1 (they never reach the heads); NOT the default")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / key-point all-gather plumbing only, on the CPU with gloo (no GPU work, "
                         "no timing claim): what tests/test_distributed.py runs in the build container")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))     # parent: never touches the GPU
    if args.dry_run:
        return dry_run(args)
    if args.mode == "stream":
        return stream_mode(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.distributed import gather_keypoints
    from sgtapose_amd.synthetic import make_inputs
    B, H, W = args.batch, args.height, args.width
    seq = args.arch == "dlapawdT_34"
    three = args.arch == "dlapawd3_34" or seq
    if three:
        args.no_cpu_baseline = True                 # the CPU baseline / parity leg is the two-frame oracle's
    if seq:
        from sgtapose_amd.synthetic import SEQ_ARGS, make_inputs_seq
        inp = make_inputs_seq(B, H, W, T=args.frames, seed=317 + rank * B)
        dev_in = [[u.to(dev) for u in inp[k]] if isinstance(inp[k], list) else inp[k].to(dev) for k in SEQ_ARGS]
    elif three:
        from sgtapose_amd.synthetic import THREE_ARGS, make_inputs_three
        inp = make_inputs_three(B, H, W, seed=317 + rank * B)
        dev_in = [inp[k].to(dev) for k in THREE_ARGS]
    else:
        inp = make_inputs(B, H, W, seed=317 + rank * B)        # this rank's shard of the global batch
        dev_in = [inp[k].to(dev) for k in ARGS]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def workload(dtype):
        """Model + detector of one compute dtype and its step functions."""
        model, sd = build_model(dev, arch=args.arch, frames=args.frames)
        model.engine_dtype = dtype
        eng = model.engine()
        eng.compute_dead_levels = not args.skip_dead_levels
        if args.early_levels is not None:
            eng.early_levels = tuple(int(v) for v in args.early_levels.split(",") if v != "")
        det = BatchedDetector(model)
        state = {}

        def local_step():
            out, dets = det.process_device(*dev_in, static=True)
            state["out"], state["dets"], state["kp"] = out, dets, det.keypoints(dets)

        def run_step():
            local_step()
            # the path's only exchange step: all-gather of the [B,7,3] keypoints (RCCL when world > 1)
            state["kp_all"] = gather_keypoints(state["kp"], world * B)

        return dict(model=model, sd=sd, eng=eng, det=det, state=state, local_step=local_step, run_step=run_step)

    wk = workload(args.dtype)
    state = wk["state"]
    el, enq = timed_steps(wk["run_step"], barrier, args.steps, args.warmup)
    ranks_seen, kp_sums_equal = 1, True
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        ranks_seen = dist.get_world_size()
        assert tuple(state["kp_all"].shape) == (world * B, 7, 3)
        # outside the timed region: every rank must hold the SAME gathered key-point tensor
        cs = state["kp_all"].double().nan_to_num().sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        kp_sums_equal = bool(lo.item() == hi.item())
        if rank == 0 and (ranks_seen != world or not kp_sums_equal):
            raise SystemExit("multi-rank check failed: ranks_seen %d of %d, gathered key-points equal on all ranks: %s"
                             % (ranks_seen, world, kp_sums_equal))

    unit_name = ("frame-windows(T=%d)" % args.frames) if seq else ("frame-triples" if three else "frame-pairs")
    cfg_name = {(32, 480, 640, "f32"): "configs[1]", (32, 480, 640, "bf16"): "configs[2] per-GPU shard (32 pairs, bf16)",
                (32, 480, 640, "f16"): "configs[2] per-GPU shard on IEEE half (32 pairs, f16)",
                (32, 480, 640, "f32x3"): "configs[1] at fp32 accuracy, convolutions emulated on the 16-bit matrix pipe (f32x3)",
                (16, 736, 1280, "f32"): "configs[3] (720p padded to a multiple of 32)",
                (16, 736, 1280, "f32x3"): "configs[3] (720p padded to a multiple of 32) at fp32 accuracy on the 16-bit matrix pipe (f32x3)",
                (16, 736, 1280, "bf16"): "configs[3] (720p padded to a multiple of 32) on the bf16 data path",
                (16, 736, 1280, "f16"): "configs[3] (720p padded to a multiple of 32) on IEEE half"}.get((B, H, W, args.dtype), "custom")
    if three:
        cfg_name = "three-frame variant dlapawd3_34 (seq_len 3: the reference's widest; towards configs[4])"
    if seq:
        cfg_name = "configs[4] per-GPU shard: temporal window of %d frames (prev-%d + curr), dlapawdT_34" % (args.frames, args.frames - 1)
    what = ("%d-frame windows" % args.frames) if seq else ("3-frame triples" if three else "2-frame pairs")
    result = {
        "metric": "%s/sec @%dx%d" % (unit_name, args.height, args.width),
        "value": round(world * B * args.steps / el, 3),
        "unit": "frame-windows/s" if seq else ("frame-triples/s" if three else "frame-pairs/s"),
        "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * el / args.steps, 3),
        "host_enqueue_ms_per_step": round(1e3 * enq / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_LABEL.get(args.dtype, args.dtype), "data": "synthetic",
        "config": {"workload": cfg_name + ": batch=%d synthetic %s of %dx%d per GPU, %s random-init (seeded), "
                               "full backbone + temporal attention (6 levels) + DCN neck + heads + sigmoid + peak "
                               "decode%s" % (B, what, H, W, args.arch, " + RCCL all-gather of keypoints" if world > 1 else ""),
                   "batch_per_gpu": B, "global_batch": world * B, "net_input": [H, W],
                   "parallelism": "dp%d (batch shards, weights replicated)" % world,
                   "dead_levels_computed": not args.skip_dead_levels,
                   "workspace_GB": round(wk["eng"].workspace_bytes() / 2 ** 30, 2)},
    }
    if world > 1:
        result["gathered_keypoints_equal_on_all_ranks"] = kp_sums_equal
    if args.sustained_steps > 0:
        # a longer region right behind the contract's K steps (clocks settled, no warm-up effects): the sustained rate
        el2, _ = timed_steps(wk["run_step"], barrier, args.sustained_steps, 0)
        if dist is not None:
            tt = torch.tensor([el2], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el2 = float(tt.item())
        result["sustained_value"] = round(world * B * args.sustained_steps / el2, 3)
        result["sustained_steps"] = args.sustained_steps

    # everything below is rank-0-only and collective-free (the other ranks wait in the final barrier)
    wl = {"batch": B, "height": H, "width": W, "arch": args.arch, "dtype": args.dtype}
    if rank == 0 and not args.no_roofline:
        result["roofline"], result["kernels"], result["other_kernels_ms"] = roofline_record(
            wk["local_step"], args.dtype, wl, 1e3 * el / args.steps, args.breakdown)

    refs = prefs = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        wk["local_step"]()
        torch.cuda.synchronize()
        cb, refs = cpu_reference(wk["sd"], inp)
        cb["parity"] = batch_parity(refs, state["out"], state["dets"], args.dtype)
        # samples of the seeded configs[1] batch in which the reference finds key-points with the peaked weight set
        # (scanned offline with the oracle: 3 / 2 / 2 key-points); any other workload: first / middle / last
        picks = (7, 9, 30) if (B, H, W, rank) == (32, 480, 640, 0) else sorted(refs)[::2][:3]
        prefs = peaked_reference(inp, picks)
        cb["parity"]["peaked_weights"] = peaked_parity(dev, inp, prefs, args.dtype)
        result["cpu_baseline"] = cb
        result["parity_ok"] = bool(cb["parity"]["within_bar"] and (args.dtype not in ("f32", "f32x3") or (
            cb["parity"]["integer_peaks_identical"] and cb["parity"]["peaked_weights"]["integer_peaks_identical"])))

    # ---- further compute dtypes of the SAME workload, timed in this process with the same K / W and bracket ---------------
    sec = args.secondary
    if sec is None:
        sec = "f32x3,bf16,f16" if (world == 1 and args.dtype == "f32" and not three and (B, H, W) == (32, 480, 640)) else ""
    sec = [d for d in sec.split(",") if d and d != args.dtype]
    if rank == 0 and world == 1 and sec:
        wk["eng"].bufs.clear()
        del wk
        torch.cuda.empty_cache()
        result["secondary"] = {}
        for dt in sec:
            w2 = workload(dt)
            el_s, enq_s = timed_steps(w2["run_step"], barrier, args.steps, args.warmup)
            rec = {"dtype": DTYPE_LABEL.get(dt, dt), "value": round(B * args.steps / el_s, 3), "unit": result["unit"], "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": round(1e3 * el_s / args.steps, 3),
                   "host_enqueue_ms_per_step": round(1e3 * enq_s / args.steps, 3),
                   "workload": "the primary workload (batch=%d, %dx%d, %s) on the %s data path" % (B, H, W, args.arch, dt)}
            if args.sustained_steps > 0:
                el2, _ = timed_steps(w2["run_step"], barrier, args.sustained_steps, 0)
                rec["sustained_value"] = round(B * args.sustained_steps / el2, 3)
            if not args.no_roofline:
                rec["roofline"], rec["kernels"], _ = roofline_record(w2["local_step"], dt, dict(wl, dtype=dt),
                                                                      1e3 * el_s / args.steps, args.breakdown)
            if refs is not None:
                w2["local_step"]()
                torch.cuda.synchronize()
                rec["parity"] = batch_parity(refs, w2["state"]["out"], w2["state"]["dets"], dt)
                rec["parity"]["peaked_weights"] = peaked_parity(dev, inp, prefs, dt)
                rec["parity_ok"] = bool(rec["parity"]["within_bar"] and (dt != "f32x3" or (
                    rec["parity"]["integer_peaks_identical"] and rec["parity"]["peaked_weights"]["integer_peaks_identical"])))
            result["secondary"][dt] = rec
            w2["eng"].bufs.clear()
            del w2
            torch.cuda.empty_cache()

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and result.get("parity_ok") is False:
        raise SystemExit(4)                         # a step whose outputs leave the dtype's bar is not a valid measurement


def stream_mode(args):
    """--mode stream: the reference's own operating point (SGTADetector.run, sgta_detector.py:117-236 under inference_real,
    inference.py:209-222): B cameras (default 1) in lock-step, frame after frame, the whole loop on the device
    (sgtapose_amd.stream.VideoRunner).  Per compute dtype, eager launches and HIP-graph replay:
      latency_ms    host-observed time of one frame, synchronised every frame: H2D of the raw uint8 frame + the step +
                    D2H of the [B,7,2] key-point table (median / p90 over --steps frames)
      frames_per_s  the same loop enqueued without per-frame synchronisation (the loop's frame-to-frame dependency lives
                    on the device, so the host may run ahead)
      host_enqueue_ms  host time to enqueue one frame (what bounds the eager form at small batch)."""
    import numpy as np
    from sgtapose_amd.stream import VideoRunner
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    B, H, W = (args.batch if args.batch != 32 else 1), args.height, args.width
    steps, warm = max(args.steps, 20), max(args.warmup, 6)
    g = torch.Generator().manual_seed(317)
    frames = [torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8).pin_memory() for _ in range(8)]
    pts = np.array([[0, 0, 0], [0.1, 0.05, 0.3], [-0.15, 0.1, 0.25], [0.2, -0.1, 0.1], [-0.1, -0.15, 0.2], [0.05, 0.2, 0.05],
                    [0.25, 0.15, 0.35]], np.float64)
    x3d = np.stack([pts] * B)
    camK = np.array([[0.94 * W, 0, W / 2.0], [0, 0.94 * W, H / 2.0], [0, 0, 1.0]])
    dtypes = [args.dtype] + [d for d in (args.secondary if args.secondary is not None else "f32x3,bf16,f16").split(",")
                             if d and d != args.dtype]
    rec = {}
    for dt in dtypes:
        from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt
        from util import golden_state_dict
        model = create_model("dlapawdl3new_34", DEFAULT_HEADS, default_head_conv(),
                             default_opt(input_h=H, input_w=W, fix_res=True))
        model.load_state_dict(golden_state_dict(0, peaked=True), strict=True)
        model = model.to(dev).eval()
        model.engine_dtype = dt
        rec[dt] = {}
        for use_graph in (False, True):
            r = VideoRunner(model, (H, W), camK, batch=B, use_graph=use_graph)
            for i in range(warm):
                r.step(frames[i % 8], x3d)
            lat = []
            for i in range(steps):
                t0 = time.perf_counter()
                r.step(frames[i % 8], x3d)
                lat.append(1e3 * (time.perf_counter() - t0))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                r.step(frames[i % 8], x3d, sync=False)
            enq = time.perf_counter() - t0
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            lat.sort()
            rec[dt]["graph" if use_graph else "eager"] = {
                "latency_ms_median": round(_median(lat), 3), "latency_ms_p90": round(lat[int(0.9 * (len(lat) - 1))], 3),
                "frames_per_s": round(B * steps / el, 1), "host_enqueue_ms": round(1e3 * enq / steps, 3)}
            log("stream %-4s %-5s latency %.2f ms (p90 %.2f)  pipelined %.0f frames/s  host enqueue %.2f ms/frame"
                % (dt, "graph" if use_graph else "eager", _median(lat), lat[int(0.9 * (len(lat) - 1))], B * steps / el,
                   1e3 * enq / steps))
            del r
        model.engine().bufs.clear()
        del model
        torch.cuda.empty_cache()
    best = rec[args.dtype]["graph"]
    print(json.dumps({
        "metric": "ms/frame, closed-loop video at %dx%d, %d camera(s): raw uint8 frame in -> key-point table out" % (H, W, B),
        "value": best["latency_ms_median"], "unit": "ms", "n_gpus": 1, "steps": steps, "warmup": warm,
        "ms_per_step": best["latency_ms_median"], "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_LABEL.get(args.dtype, args.dtype), "data": "synthetic",
        "config": {"workload": "stream mode (the reference's SGTADetector.run loop): pre-process -> PnP priors -> two-frame "
                               "network -> decode -> post-process per frame, state on the device; HIP-graph replay",
                   "cameras": B, "net_input": [H, W]},
        "stream": rec}), flush=True)


if __name__ == "__main__":
    main()
