"""Which side's 128-B misalignment costs what?  Shifts the ASCII and packed buffers independently.

usage (GPU box): python bench/align_lab.py [--log2-nt 32]   -> gpurun_out/align_lab.json
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import devutil  # noqa: E402


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-nt", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--offsets", default="0,8,16,32,64,1,5,100")
    ap.add_argument("--packed-offsets", default="")
    ap.add_argument("--five", action="store_true", help="the 5-letter codec instead of the 2-bit one")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "align_lab.json"))
    args = ap.parse_args()
    n = 1 << args.log2_nt
    if args.five:
        n = n // 3456 * 3456
    per, bpn = (27, 1 + 8 / 27) if args.five else (32, 1.25)
    enc = cn.n_to_bits2_dev if args.five else cn.n_to_bits_dev
    dec = cn.bits_to_n2_dev if args.five else cn.bits_to_n_dev
    dev = torch.device("cuda", 0)
    pad = 16384
    b_in = torch.empty(n + pad, dtype=torch.uint8, device=dev)
    b_out = torch.empty(n + pad, dtype=torch.uint8, device=dev)
    b_pk = torch.empty(n // per + pad // 8, dtype=torch.int64, device=dev)
    offs = [int(x) for x in args.offsets.split(",")]
    poffs = [int(x) for x in args.packed_offsets.split(",")] if args.packed_offsets else None
    rows = []
    for a_off in offs:            # ASCII side (encode input / decode output)
        for p_off in (poffs or offs):  # packed side, bytes; must be a multiple of 8
            if p_off % 8:
                continue
            d_in = b_in[a_off:a_off + n]
            d_out = b_out[a_off:a_off + n]
            d_pk = b_pk[p_off // 8:p_off // 8 + n // per]
            (devutil.fill_random_acgtn if args.five else devutil.fill_random_acgt)(b_in[:n], 0x5EED)
            d_in.copy_(b_in[:n].clone())
            enc(d_in, out=d_pk)
            dec(d_pk, n, out=d_out)
            assert devutil.count_mismatch(d_in.contiguous(), d_out.contiguous()) == 0 if a_off % 16 == 0 else bool((d_in == d_out).all())
            te, td = [], []
            for _ in range(args.rounds):
                te.append(timed(lambda: enc(d_in, out=d_pk), args.iters))
                td.append(timed(lambda: dec(d_pk, n, out=d_out), args.iters))
            r = {"ascii_off": a_off, "packed_off": p_off,
                 "encode_GBs": round(bpn * n / statistics.median(te) / 1e6, 1),
                 "decode_GBs": round(bpn * n / statistics.median(td) / 1e6, 1)}
            print(json.dumps(r), flush=True)
            rows.append(r)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({"log2_nt": args.log2_nt, "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
