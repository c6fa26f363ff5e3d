"""PMC workload: the encode kernel at several residency caps / maps (variants 0, 9, 11, 5, 10)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cute_nucleotides_amd as cn  # noqa: E402
from cute_nucleotides_amd import devutil  # noqa: E402
from cute_nucleotides_amd import _lib as _cnt_lib  # noqa: E402

_cnt_lib.use_lab_build()  # this script selects kernel variants: bench/libcute_nt_hip_lab.so, not the product library


n = 1 << 34
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_pk = torch.empty(n // 32, dtype=torch.int64, device="cuda")
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
devutil.fill_random_acgt(d_in, 0x5EED)
for v in (0, 9, 11, 5, 10, 6):
    devutil.set_tuning("encode", v)
    for _ in range(3):
        cn.n_to_bits_dev(d_in, out=d_pk)
devutil.set_tuning("encode", 0)
for v in (0, 9, 10, 5):
    devutil.set_tuning("decode", v)
    for _ in range(3):
        cn.bits_to_n_dev(d_pk, n, out=d_out)
torch.cuda.synchronize()
