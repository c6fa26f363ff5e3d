"""cute_nucleotides_amd -- MI355X (gfx950) back-end for the cute-nucleotides codec hot path.

Layout:  ../hip/          HIP kernels + the C-ABI shim (-> libcute_nt_hip.so, include/cute_nt.h; `make -C hip` or build.py)
         n_to_bits.py     mirror of the reference's src/n_to_bits.rs API (2-bit codec)
         n_to_bits2.py    mirror of src/n_to_bits2.rs (5-letter codec)
         devutil.py       device-side generator / checksum / compare for benches
         sharding.py      contiguous-chunk partition used by the multi-GPU paths
         build.py         hipcc build of the library
"""
from .n_to_bits import (bits_to_n_dev, bits_to_n_hip, bits_to_n_hip_into, bits_to_n_hip_sharded, host_registered, is_pinned, n_to_bits_checked_dev,
                        n_to_bits_dev, n_to_bits_hip, n_to_bits_hip_checked, n_to_bits_hip_into, n_to_bits_hip_sharded, pinned_empty,
                        round_trip_checked_dev, round_trip_dev)
from .n_to_bits2 import (bits_to_n2_dev, bits_to_n2_hip, bits_to_n2_hip_into, n_to_bits2_checked_dev, n_to_bits2_dev, n_to_bits2_hip,
                         n_to_bits2_hip_checked, n_to_bits2_hip_into)

__all__ = [
    "n_to_bits_hip", "bits_to_n_hip", "n_to_bits_hip_sharded", "bits_to_n_hip_sharded",
    "n_to_bits_hip_into", "bits_to_n_hip_into", "n_to_bits2_hip_into", "bits_to_n2_hip_into",
    "n_to_bits_dev", "bits_to_n_dev", "round_trip_dev", "n_to_bits2_hip", "bits_to_n2_hip", "n_to_bits2_dev", "bits_to_n2_dev",
    "n_to_bits_hip_checked", "n_to_bits2_hip_checked", "n_to_bits_checked_dev", "n_to_bits2_checked_dev", "round_trip_checked_dev",
    "pinned_empty", "is_pinned", "host_registered",
]
