"""VERDICT r5 #4 (second half): would a stream-K split of the N = 768 GEMMs of a PACKED step pay?  At M' ~ 8 800 rows the 256x128 tiles of an
N = 768 launch number 35 x 6 = 210 on 256 CUs; a stream-K schedule would hand every CU 210 * (K / 64) / 256 k tiles instead of K / 64.  This probe
measures, with cold operands (12 operand sets in rotation, as the 12 layers of a step), on the product kernel (variant 77):
  (a) the launch as it is:                      M', N = 768, K
  (b) the no-seam BOUND of the balanced schedule: the same launch with K' = 64 * ceil(210 * (K / 64) / 256) (every workgroup walks exactly the k tiles
      a balanced CU would; the result is meaningless, the time is the bound),
  (c) the seam traffic: 256 CUs -> ~255 tile seams; each seam hands a 256 x 128 fp32 partial tile (128 KB) from one workgroup to another: written once,
      read once, summed with the owner's accumulators.  Timed here as its memory traffic alone: one fp32 read-add-write pass over 255 x 128 KB.
gain <= (a) - (b) - (c) per launch."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K  # noqa: E402

DEV = torch.device("cuda:0")
ROT = 12


def timeit(fn, iters=48, warm=12):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    g = torch.Generator(device=DEV)
    g.manual_seed(0)

    def r(*s, scale=0.5):
        return (torch.randn(*s, device=DEV, generator=g) * scale).half()

    out = {}
    for M in (8800, 8832, 10688):
        tiles = -(-M // 256) * 6
        for Kd in (3072, 2304, 768):
            kt = Kd // 64
            kb = -(-tiles * kt // 256)
            row = {"tiles": tiles, "k_tiles_per_workgroup_now": kt, "k_tiles_per_cu_balanced": kb}
            for tag, kk in (("as_is_us", Kd), ("balanced_no_seam_bound_us", 64 * kb)):
                sets = [(r(M, kk), r(768, kk, scale=0.05), torch.empty(M, 768, device=DEV, dtype=torch.half)) for _ in range(ROT)]
                ctr = [0]

                def call():
                    x, w, y = sets[ctr[0] % ROT]
                    ctr[0] += 1
                    K.gemm_nt(x, w, y, M, 768, kk, variant=77)
                row[tag] = round(timeit(call), 2)
                del sets
            seams = min(255, tiles)
            n = seams * 256 * 128
            bufs = [(torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)) for _ in range(ROT)]
            ctr = [0]

            def seam():
                a, b = bufs[ctr[0] % ROT]
                ctr[0] += 1
                a.add_(b)                      # read partial + read accumulator image + write: the traffic of the hand-off (upper bound: accumulators live in registers)
            row["seam_traffic_us"] = round(timeit(seam), 2)
            del bufs
            row["gain_bound_us"] = round(row["as_is_us"] - row["balanced_no_seam_bound_us"] - row["seam_traffic_us"], 2)
            out["M=%d K=%d" % (M, Kd)] = row
            print("M=%5d K=%4d: %d tiles, %2d -> %2d k tiles: as is %6.2f us, balanced bound %6.2f us, seam traffic %5.2f us => gain <= %5.2f us" % (
                M, Kd, tiles, kt, kb, row["as_is_us"], row["balanced_no_seam_bound_us"], row["seam_traffic_us"], row["gain_bound_us"]))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/streamk_bound_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
