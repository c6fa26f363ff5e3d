//! `*_hip` back-end for `cute_nucleotides` -- thin wrappers over the C ABI of
//! `libcute_nt_hip.so` (include/cute_nt.h).  Drop this file next to `n_to_bits.rs` /
//! `n_to_bits2.rs`, add `pub mod hip;` to `src/lib.rs`, and link with
//! `cargo:rustc-link-lib=dylib=cute_nt_hip` (see `build.rs`).
//!
//! NOT COMPILED in this repository's CI: the build image has no Rust toolchain.  The
//! wrappers are deliberately mechanical -- allocate a `Vec` of the exact size, call the C
//! symbol, `set_len`, return -- so that review can stand in for compilation.
//!
//! Signatures are identical to the reference's (`n_to_bits.rs:34,51`, `n_to_bits2.rs:37,78`).

use std::os::raw::{c_char, c_int, c_uint};

#[link(name = "cute_nt_hip")]
extern "C" {
    fn cnt_strerror(status: c_int) -> *const c_char;
    fn cnt_words_for(n_len: usize) -> usize;
    fn cnt_words2_for(n_len: usize) -> usize;
    fn cnt_n_to_bits_ex(n: *const u8, n_len: usize, out: *mut u64, out_words: usize, flags: c_uint) -> c_int;
    fn cnt_bits_to_n(bits: *const u64, words: usize, len: usize, out: *mut u8) -> c_int;
    fn cnt_n_to_bits2(n: *const u8, n_len: usize, out: *mut u64, out_words: usize) -> c_int;
    fn cnt_bits_to_n2(bits: *const u64, words: usize, len: usize, out: *mut u8) -> c_int;
    fn cnt_n_to_bits_sharded(n: *const u8, n_len: usize, out: *mut u64, out_words: usize, ndev: c_int) -> c_int;
    fn cnt_bits_to_n_sharded(bits: *const u64, words: usize, len: usize, out: *mut u8, ndev: c_int) -> c_int;
    fn cnt_n_to_bits2_sharded(n: *const u8, n_len: usize, out: *mut u64, out_words: usize, ndev: c_int) -> c_int;
    fn cnt_bits_to_n2_sharded(bits: *const u64, words: usize, len: usize, out: *mut u8, ndev: c_int) -> c_int;
}

const CNT_STRICT_LUT: c_uint = 1;

fn check(status: c_int) {
    if status != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(cnt_strerror(status)) };
        // CNT_ELEN carries the reference's own panic text (n_to_bits.rs:53)
        panic!("{}", msg.to_string_lossy());
    }
}

/// Encode `{A, T/U, C, G}` into pairs of bits packed into 64-bit integers on the GPU.
/// Same contract as `n_to_bits_lut` (`n_to_bits.rs:34`) on the nucleotide alphabet.
pub fn n_to_bits_hip(n: &[u8]) -> Vec<u64> {
    let words = unsafe { cnt_words_for(n.len()) };
    let mut res: Vec<u64> = Vec::with_capacity(words);
    unsafe {
        check(cnt_n_to_bits_ex(n.as_ptr(), n.len(), res.as_mut_ptr(), words, 0));
        res.set_len(words);
    }
    res
}

/// As `n_to_bits_hip`, but bytes outside `ACGTUacgtu` encode as 0 like `BYTE_LUT` does.
pub fn n_to_bits_hip_strict(n: &[u8]) -> Vec<u64> {
    let words = unsafe { cnt_words_for(n.len()) };
    let mut res: Vec<u64> = Vec::with_capacity(words);
    unsafe {
        check(cnt_n_to_bits_ex(n.as_ptr(), n.len(), res.as_mut_ptr(), words, CNT_STRICT_LUT));
        res.set_len(words);
    }
    res
}

/// Decode pairs of bits from packed 64-bit integers on the GPU (`n_to_bits.rs:51`).
pub fn bits_to_n_hip(bits: &[u64], len: usize) -> Vec<u8> {
    if len > (bits.len() << 5) {
        panic!("The length is greater than the number of nucleotides!");
    }
    let mut res: Vec<u8> = Vec::with_capacity(len);
    unsafe {
        check(cnt_bits_to_n(bits.as_ptr(), bits.len(), len, res.as_mut_ptr()));
        res.set_len(len);
    }
    res
}

/// 5-letter codec (`n_to_bits2.rs:37`).
pub fn n_to_bits2_hip(n: &[u8]) -> Vec<u64> {
    let words = unsafe { cnt_words2_for(n.len()) };
    let mut res: Vec<u64> = Vec::with_capacity(words);
    unsafe {
        check(cnt_n_to_bits2(n.as_ptr(), n.len(), res.as_mut_ptr(), words));
        res.set_len(words);
    }
    res
}

/// 5-letter codec (`n_to_bits2.rs:78`).
pub fn bits_to_n2_hip(bits: &[u64], len: usize) -> Vec<u8> {
    if len > bits.len() * 27 {
        panic!("The length is greater than the number of nucleotides!");
    }
    let mut res: Vec<u8> = Vec::with_capacity(len);
    unsafe {
        check(cnt_bits_to_n2(bits.as_ptr(), bits.len(), len, res.as_mut_ptr()));
        res.set_len(len);
    }
    res
}

/// Contiguous-chunk sharding over `ndev` GPUs (0 = all visible); no collective.
pub fn n_to_bits_hip_sharded(n: &[u8], ndev: i32) -> Vec<u64> {
    let words = unsafe { cnt_words_for(n.len()) };
    let mut res: Vec<u64> = Vec::with_capacity(words);
    unsafe {
        check(cnt_n_to_bits_sharded(n.as_ptr(), n.len(), res.as_mut_ptr(), words, ndev));
        res.set_len(words);
    }
    res
}

pub fn bits_to_n_hip_sharded(bits: &[u64], len: usize, ndev: i32) -> Vec<u8> {
    if len > (bits.len() << 5) {
        panic!("The length is greater than the number of nucleotides!");
    }
    let mut res: Vec<u8> = Vec::with_capacity(len);
    unsafe {
        check(cnt_bits_to_n_sharded(bits.as_ptr(), bits.len(), len, res.as_mut_ptr(), ndev));
        res.set_len(len);
    }
    res
}

/// The 5-letter codec over `ndev` GPUs.
pub fn n_to_bits2_hip_sharded(n: &[u8], ndev: i32) -> Vec<u64> {
    let words = unsafe { cnt_words2_for(n.len()) };
    let mut res: Vec<u64> = Vec::with_capacity(words);
    unsafe {
        check(cnt_n_to_bits2_sharded(n.as_ptr(), n.len(), res.as_mut_ptr(), words, ndev));
        res.set_len(words);
    }
    res
}

pub fn bits_to_n2_hip_sharded(bits: &[u64], len: usize, ndev: i32) -> Vec<u8> {
    if len > bits.len() * 27 {
        panic!("The length is greater than the number of nucleotides!");
    }
    let mut res: Vec<u8> = Vec::with_capacity(len);
    unsafe {
        check(cnt_bits_to_n2_sharded(bits.as_ptr(), bits.len(), len, res.as_mut_ptr(), ndev));
        res.set_len(len);
    }
    res
}

#[cfg(test)]
mod tests {
    use super::*;

    // the reference's own vectors (n_to_bits.rs:412-423, n_to_bits2.rs:274-285)
    #[test]
    fn test_n_to_bits_hip() {
        assert_eq!(n_to_bits_hip(b"ATCGATCGATCGATCGATCGATCGATCGATCG"), vec![0xD8D8D8D8D8D8D8D8u64]);
        assert_eq!(n_to_bits_hip(b"ATCG"), vec![0b11011000]);
    }

    #[test]
    fn test_bits_to_n_hip() {
        assert_eq!(bits_to_n_hip(&vec![0xD8D8D8D8D8D8D8D8u64], 32), "ATCGATCGATCGATCGATCGATCGATCGATCG".as_bytes());
    }

    #[test]
    fn test_n_to_bits2_hip() {
        assert_eq!(n_to_bits2_hip(b"ATCGNATCGNATCGNATCGNATCGNATCGNATCGN"), vec![0x36A45D1F46D48BA3u64, 0x5D1F4]);
        assert_eq!(n_to_bits2_hip(b"ATCGN"), vec![0b101110100011]);
    }

    #[test]
    fn test_bits_to_n2_hip() {
        assert_eq!(bits_to_n2_hip(&vec![0x36A45D1F46D48BA3u64, 0x5D1F4], 35), "ATCGNATCGNATCGNATCGNATCGNATCGNATCGN".as_bytes());
    }
}
