#!/usr/bin/env python
"""Per-shape table of the conv forward / input-gradient / weight-gradient calls inside the C2 train step (the launches bench.py's `roofline`
and `roofline_wgrad` average over): label, shape, calls per step, average duration, TFLOP/s, share of the step's conv time.
usage: python tools/step_shapes.py [steps]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd import ops
from dmvae_amd.train import build_tokenizer_trainer


class ShapeLog(list):
    """ops.* append (label, e0, e1, flop); the shape is read from the caller's frame."""
    def append(self, item):
        f = sys._getframe(1).f_locals
        if item[0] == "gemm_pp_kernel":       # ops.linear_bf16: the encoder's Linear layers
            list.append(self, ("gemm_pp", "%dx%dx%d%s" % (f["m"], f["n"], f["k"], " act%d" % f["act"] if f.get("act") else "")) + tuple(item[1:]))
            return
        shape = "%dx%dx%d %d>%d k%d%s%s%s" % (f["n"], f.get("ho", 0) or f["dy"].shape[1], f.get("wo", 0) or f["dy"].shape[2], f["cin"], f["cout"], f["ks"],
                                            " ups%d" % int(f["upsample"]) if f.get("upsample") else "", " s%d" % f["stride"] if f.get("stride", 1) != 1 else "",
                                            " T" if f.get("transposed") else "")
        list.append(self, (item[0].replace("conv_pp_kernel", "pp").replace(", 2, 4, 4", ""), shape) + tuple(item[1:]))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda", 0)
    tr = build_tokenizer_trainer(device=dev, seed=42)
    images = torch.rand(32, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(42)) * 2 - 1
    for _ in range(3):
        tr.step(images)
    log = ShapeLog()
    ops.KERNEL_TIMING = log
    for _ in range(steps):
        tr.step(images)
    ops.KERNEL_TIMING = None
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for label, shape, e0, e1, fl in log:
        a = agg.setdefault((label, shape), [0.0, 0.0, 0])
        a[0] += e0.elapsed_time(e1); a[1] += fl; a[2] += 1
    tot = sum(a[0] for a in agg.values())
    print(f"# {steps} steps, conv + wgrad calls {tot / steps:.2f} ms/step")
    for (label, shape), (ms, fl, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"{ms / steps:7.3f} ms/step {n // steps:3d}x {ms / n * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s  {shape:34s} {label}")


if __name__ == "__main__":
    main()
