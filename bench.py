#!/usr/bin/env python3
"""Benchmark of the SGTAPose inference hot path on MI355X (contract: see the task's bench rules).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--height H] [--width W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

One "step" = one pass of the WHOLE hot path over one batch of synthetic frame pairs per GPU:
two-frame DLA-34 backbone, temporal cross-attention (all six levels), DCN neck, heads with fused
sigmoid, device decode (peaks + gathers), and -- for N > 1 -- the RCCL all-gather of the [B,7,3]
keypoints.  Inputs are resident in HBM before the timed region.  Workload at N=1: BASELINE.json
configs[1] (batch 32, 2-frame 480x640, fp32); the same per-GPU batch for N>1 (weak scaling, the
batch shards over ranks with no data-path collective except the keypoint all-gather).

After the timed region the same process (a) re-runs a few steps with HIP events around every C-ABI
launch (on the launch stream) to get the per-kernel durations the ``roofline`` object is computed
from, and (b) on rank 0 at N=1 times the CPU oracle (our restatement of the reference path; the
reference's Python cannot travel to the GPU box) on a bounded sample for ``cpu_baseline``.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import torch  # noqa: E402

ARGS = ("x", "pre_img", "pre_hm", "repro_hm", "pre_hm_cls", "repro_hm_cls")
PEAK_TFLOPS_F32_MFMA = 157.3        # MI355X_MICROARCH.md: f32-input MFMA = 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0               # HBM3E spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_model(dev, seed=0, arch="dlapawdl3new_34"):
    from util import golden_state_dict
    from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt
    if arch == "dlapawd3_34":                     # three-frame variant
        heads = {"hm": 7, "reg": 2, "wh": 2, "tracking": 2}
        sd = golden_state_dict(seed, keys="model_keys_three.json")
        model = create_model(arch, heads, default_head_conv(heads), default_opt())
    else:
        sd = golden_state_dict(seed)
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
            a = agg.setdefault(key, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0, exec_flops=0.0))
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


def pmc_traffic(kernel):
    """HBM bytes per launch of ``kernel`` from the committed PMC passes (profiles/r01_pmc_step_by_kernel.csv:
    separate ``rocprofv3 --pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` runs of this same command, see profiles/README.md),
    with the guide's gfx950 correction (FETCH_SIZE tallies the 128-byte requests of 16-B/lane streams -- the window
    and weight LDS-DMA of this kernel -- at 64 bytes: doubled).  bench.py cannot collect PMCs itself; None if the
    table or the kernel is missing."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_step_by_kernel.csv")
    try:
        import csv
        for r in csv.DictReader(open(path)):
            if r["kernel"] == kernel:
                n = int(r["launches"])
                byts = (2.0 * float(r["FETCH_SIZE_MB_raw"]) + float(r["WRITE_SIZE_MB_raw"])) * 1e6 / n
                return round(byts), ("bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) / %d launches of one step, "
                                     "profiles/r01_pmc_step_by_kernel.csv (separate --pmc passes; not collected live)" % n)
    except (OSError, KeyError, ValueError):
        pass
    return None, "no PMC table for this kernel under profiles/"


def cpu_baseline(sd, H, W, det=None, dev=None, budget_s=20.0):
    """The CPU oracle (kind 'port') on B=1 frame pairs of the same workload, all host cores; the same sample then
    goes through the HIP path and the heat-map / key-point deviation is reported beside the timing ('parity')."""
    import numpy as np
    import oracle
    from sgtapose_amd.synthetic import make_inputs
    cores = torch.get_num_threads()
    inp = make_inputs(1, H, W, seed=999)
    a = [inp[k] for k in ARGS]
    ref_out, ref_dets = oracle.process_reference(sd, *a)   # warm-up (and the parity reference)
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.process_reference(sd, *a)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 16:
            break
    res = {"value": round(n / el, 4), "unit": "frame-pairs/s", "cores": cores, "kind": "port",
           "sample": "%d x (B=1, %dx%d) frame pairs through oracle.process_reference (forward + sigmoid + "
                     "decode), %d torch threads, %.1f s" % (n, H, W, cores, el)}
    if det is not None:
        out, dets = det.process_device(*[t.to(dev) for t in a])
        kp_hip = det.keypoints(dets)[0].cpu().numpy().astype(np.float64)           # [7, 3] = x, y, score
        kp_ref = np.concatenate([ref_dets["cts_wreg"][0].reshape(7, 2), ref_dets["scores"][0].reshape(7, 1)], 1)
        found = kp_ref[:, 2] >= 0
        res["parity"] = {
            "hm_max_abs": float((out["hm"].cpu() - ref_out["hm"]).abs().max()),
            "reg_max_abs": float((out["reg"].cpu() - ref_out["reg"]).abs().max()),
            "tracking_max_abs": float((out["tracking"].cpu() - ref_out["tracking"]).abs().max()),
            "keypoints_found_ref": int(found.sum()),
            "keypoints_found_same": bool(np.array_equal(found, kp_hip[:, 2] >= 0)),
            "keypoint_l2_max_px": float(np.sqrt(((kp_hip[found, :2] - kp_ref[found, :2]) ** 2).sum(1)).max()) if found.any() else 0.0,
            "bar": "1e-3 abs (north_star); same seeded sample through the CPU oracle and the HIP path"}
        # random-init logits are flat (no peaks): key-point parity of the decode on peaked maps fed to both decoders
        from sgtapose_amd.decode import dream_generic_decode
        from sgtapose_amd.synthetic import peaked_heatmaps
        hm, _ = peaked_heatmaps(4, H // 4, W // 4, seed=7)
        reg, trk = ref_out["reg"].repeat(4, 1, 1, 1), ref_out["tracking"].repeat(4, 1, 1, 1)
        got = dream_generic_decode({"hm": hm.to(dev), "reg": reg.to(dev), "tracking": trk.to(dev)}, K=7)
        ref = oracle.dream_decode({"hm": hm, "reg": reg, "tracking": trk})
        d = (got["cts_wreg"].cpu().double() - ref["cts_wreg"].double()).reshape(-1, 2)
        res["parity"]["decode_peaked_maps"] = {
            "keypoints": int((ref["scores"] > 0).sum()),
            "integer_peaks_identical": bool(torch.equal(got["xs"].cpu(), ref["xs"]) and torch.equal(got["ys"].cpu(), ref["ys"])),
            "keypoint_l2_max_px": float(d.pow(2).sum(1).sqrt().max())}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="frame pairs per GPU per step")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--arch", default="dlapawdl3new_34", choices=["dlapawdl3new_34", "dlapawd3_34"],
                    help="dlapawd3_34 = the reference's three-frame variant (frame triples/s; not the headline metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--early-levels", default=None,
                    help="comma list of pyramid levels whose attention overlaps the backbone on a side stream "
                         "(engine default: 2,3; empty string: none)")
    ap.add_argument("--skip-dead-levels", action="store_true",
                    help="do not compute temporal levels 0/1 (they never reach the heads); NOT the default")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.synthetic import make_inputs
    B, H, W = args.batch, args.height, args.width
    model, sd = build_model(dev, arch=args.arch)
    three = args.arch == "dlapawd3_34"
    if three:
        args.no_cpu_baseline = True                 # the CPU baseline / parity leg is the two-frame oracle's
    model.engine().compute_dead_levels = not args.skip_dead_levels
    if args.early_levels is not None:
        model.engine().early_levels = tuple(int(v) for v in args.early_levels.split(",") if v != "")
    det = BatchedDetector(model)
    if three:
        from sgtapose_amd.synthetic import THREE_ARGS, make_inputs_three
        inp = make_inputs_three(B, H, W, seed=317 + rank * B)
        dev_in = [inp[k].to(dev) for k in THREE_ARGS]
    else:
        inp = make_inputs(B, H, W, seed=317 + rank * B)        # this rank's shard of the global batch
        dev_in = [inp[k].to(dev) for k in ARGS]
    from sgtapose_amd.distributed import gather_keypoints
    state = {}

    def run_step():
        _, dets = det.process_device(*dev_in, static=True)
        # the path's only exchange step: all-gather of the [B,7,3] keypoints (RCCL when world > 1)
        state["kp_all"] = gather_keypoints(det.keypoints(dets), world * B)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    cfg_name = {(32, 480, 640): "configs[1]", (16, 736, 1280): "configs[3] (720p padded to a multiple of 32)"}.get(
        (B, H, W), "custom")
    if three:
        cfg_name = "three-frame variant dlapawd3_34 (seq_len 3: the reference's widest; towards configs[4])"
    result = {
        "metric": "%s/sec @%dx%d" % ("frame-triples" if three else "frame-pairs", args.height, args.width),
        "value": round(world * B * args.steps / el, 3),
        "unit": "frame-triples/s" if three else "frame-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * el / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg_name + ": batch=%d 2-frame %dx%d synthetic pairs per GPU, dlapawdl3new_34 "
                               "random-init (seeded), full backbone + temporal attention (6 levels) + DCN neck + "
                               "heads + sigmoid + peak decode%s" % (B, H, W, " + RCCL all-gather of keypoints" if world > 1 else ""),
                   "batch_per_gpu": B, "global_batch": world * B, "net_input": [H, W],
                   "parallelism": "dp%d (batch shards, weights replicated)" % world,
                   "dead_levels_computed": not args.skip_dead_levels,
                   "workspace_GB": round(model.engine().workspace_bytes() / 2 ** 30, 2)},
    }

    if rank == 0 and not args.no_roofline:
        agg = instrumented_pass(run_step, 2)
        tot = sum(a["ms"] for a in agg.values())
        rows = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
        if args.breakdown:
            log("%-44s %8s %6s %9s %9s" % ("kernel (C-ABI entry / template)", "ms/step", "share", "TFLOP/s", "GB/s"))
            for k, a in rows:
                tf = a["flops"] / a["ms"] / 1e9 if a["flops"] else 0
                gb = a["bytes"] / a["ms"] / 1e6 if a["bytes"] else 0
                log("%-44s %8.3f %5.1f%% %9.1f %9.0f  x%d" % (k, a["ms"], 100 * a["ms"] / tot, tf, gb, a["launches"]))
            log("sum of kernel time %.2f ms/step (event-bracketed), timed region %.2f ms/step" % (tot, 1e3 * el / args.steps))
        top, ta = rows[0]
        ach = ta["flops"] / ta["ms"] / 1e9                     # TFLOP/s, algorithmic (un-padded) flops
        exe = ta["exec_flops"] / ta["ms"] / 1e9                # TFLOP/s actually issued to the matrix pipe
        result["roofline"] = {"kernel": top, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS_F32_MFMA,
                              "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS_F32_MFMA, 4), "traffic": None,
                              "executed": round(exe, 2), "executed_frac": round(exe / PEAK_TFLOPS_F32_MFMA, 4),
                              "note": "achieved = algorithmic (direct-conv, un-padded) FLOPs / measured kernel time; "
                                      "the Winograd kernels issue only 16/36 (F(2x2,3x3)) or 36/144 (F(4x4,3x3)) of them to "
                                      "the matrix pipe, so achieved may exceed the f32 MFMA peak while executed_frac is "
                                      "the matrix-pipe utilisation",
                              "avg_launch_ms": round(ta["ms"] / ta["launches"], 4), "launches_per_step": ta["launches"],
                              "share_of_step": round(ta["ms"] / tot, 3)}
        result["roofline"]["traffic"], result["roofline"]["traffic_note"] = pmc_traffic(top)
        extra = []
        for k, a in rows:
            if a["flops"]:
                extra.append({"kernel": k, "ms_per_step": round(a["ms"], 3),
                              "tflops": round(a["flops"] / a["ms"] / 1e9, 2),
                              "executed_tflops": round(a["exec_flops"] / a["ms"] / 1e9, 2),
                              "mfma_frac": round(a["exec_flops"] / a["ms"] / 1e9 / PEAK_TFLOPS_F32_MFMA, 4),
                              "alg_GBs": round(a["bytes"] / a["ms"] / 1e6, 1),
                              "hbm_frac": round(a["bytes"] / a["ms"] / 1e6 / PEAK_HBM_GBS, 4)})
        result["kernels"] = extra
        result["other_kernels_ms"] = {k: round(a["ms"], 3) for k, a in rows if not a["flops"]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(sd, H, W, det, dev)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
