//! MI355X (HIP) back-end for cute-nucleotides: `hip::n_to_bits_hip` & co. with the reference's
//! signatures, over the C ABI of libcute_nt_hip.so (include/cute_nt.h).  Behind the `hip` cargo feature.
#[cfg(feature = "hip")]
pub mod hip;
