#!/usr/bin/env python3
"""Per-stage error budget of the 16-bit data paths (bf16 / f16) against the fp32 HIP engine on the same inputs.

    python tools/error_budget.py [--batch 2] [--height 480] [--width 640] [--dtypes bf16,f16]

For every stage the 16-bit engine's output is compared with the fp32 engine's (which the GPU parity tests hold within
1e-5 of the CPU oracle): max / mean absolute error and the error relative to the stage's own RMS.  Two columns:

  free    the 16-bit engine end to end (errors accumulate from the input on)
  forced  the same stage fed with the fp32 engine's OUTPUT of the previous stage rounded once to the 16-bit format
          (what the stage itself adds: its operand rounding, its arithmetic, its store)

Stages: the six backbone levels (both frames), the levels after the temporal write-back, the neck output (the 64-channel
map the heads read), the three heads.  GPU only; no oracle involved.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import torch  # noqa: E402

ARGS = ("x", "pre_img", "pre_hm", "repro_hm", "pre_hm_cls", "repro_hm_cls")


def stats(got, ref):
    d = (got - ref).abs()
    rms = float(ref.pow(2).mean().sqrt())
    return float(d.max()), float(d.mean()), float(d.mean()) / max(rms, 1e-30), rms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--dtypes", default="bf16,f16")
    ap.add_argument("--seed", type=int, default=317)
    args = ap.parse_args()
    from util import golden_state_dict
    from sgtapose_amd import ops, ops_bf16 as bops
    from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt
    from sgtapose_amd.synthetic import make_inputs
    dev = torch.device("cuda", 0)
    sd = golden_state_dict(0)
    inp = make_inputs(args.batch, args.height, args.width, seed=args.seed)
    dev_in = [inp[k].to(dev) for k in ARGS]
    B = args.batch

    def model(dtype):
        m = create_model("dlapawdl3new_34", DEFAULT_HEADS, default_head_conv(), default_opt())
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).eval()
        m.engine_dtype = dtype
        e = m.engine()
        e.use_side_stream = False                   # one stream: the stages below are called one by one
        return m, e

    def run(eng, forced_from=None):
        """-> dict stage -> fp32 NCHW tensor.  forced_from = the fp32 run's stage tensors: each stage then starts from them."""
        x, pre_img, pre_hm, repro_hm, pre_cls, cur_cls = dev_in
        st = {}
        h16 = hasattr(eng, "act_dtype")
        as32 = (lambda v: v.nchw()) if h16 else (lambda v: v.base[:, v.c0:v.c0 + v.C].clone())

        def inject(view, t):                        # overwrite a whole-buffer view with a reference tensor (rounded once)
            if h16:
                view.base.copy_(bops.blocked_from_nchw(t, dtype=eng.act_dtype))
            else:
                view.base[:, view.c0:view.c0 + view.C].copy_(t)

        idx = ops.prior_argmax(pre_cls, cur_cls, eng.buf("idx", (2, B, eng.num_classes, 2)))
        lv = eng.backbone(x, pre_img, pre_hm, repro_hm)
        for i, v in enumerate(lv):
            st["backbone.l%d" % i] = as32(v)
        if forced_from is not None:
            for i, v in enumerate(lv):
                inject(v, forced_from["backbone.l%d" % i])
        for i in range(6):
            eng.temporal_scatter(lv[i], B, eng.temporal_tokens(lv[i], i, idx, B))
            st["temporal.l%d" % i] = as32(eng._frame(lv[i], 1, B))
        cur = [eng._frame(v, 1, B) for v in lv]
        if forced_from is not None:
            for i, v in enumerate(cur):
                inject(v, forced_from["temporal.l%d" % i])
        feat = eng.neck(cur)
        st["neck.feat"] = as32(feat)
        if forced_from is not None:
            inject(feat, forced_from["neck.feat"])
        out = eng.heads(feat)
        for k, v in out.items():
            st["head." + k] = v.clone()
        torch.cuda.synchronize()
        return st

    with torch.no_grad():
        _, e32 = model("f32")
        ref = run(e32)
        for dt in args.dtypes.split(","):
            _, e = model(dt)
            free = run(e)
            forced = run(e, forced_from=ref)
            print("\n== %s vs fp32 engine, B=%d %dx%d ==" % (dt, B, args.height, args.width))
            print("%-14s %10s | %-32s | %-32s" % ("stage", "ref rms", "free: max / mean / mean÷rms", "forced: max / mean / mean÷rms"))
            for k in ref:
                a, b = stats(free[k], ref[k]), stats(forced[k], ref[k])
                print("%-14s %10.4f | %10.3e %10.3e %9.2e | %10.3e %10.3e %9.2e" % (k, a[3], a[0], a[1], a[2], b[0], b[1], b[2]))


if __name__ == "__main__":
    main()
