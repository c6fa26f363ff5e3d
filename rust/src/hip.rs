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

use std::os::raw::{c_char, c_int, c_uint, c_void};

#[link(name = "cute_nt_hip")]
extern "C" {
    fn cnt_strerror(status: c_int) -> *const c_char;
    fn cnt_words_for(n_len: usize) -> usize;
    fn cnt_words2_for(n_len: usize) -> usize;
    fn cnt_n_to_bits_ex(n: *const u8, n_len: usize, out: *mut u64, out_words: usize, flags: c_uint) -> c_int;
    fn cnt_n_to_bits_checked(n: *const u8, n_len: usize, out: *mut u64, out_words: usize, flags: c_uint, invalid: *mut u64) -> c_int;
    fn cnt_n_to_bits2_checked(n: *const u8, n_len: usize, out: *mut u64, out_words: usize, flags: c_uint, invalid: *mut u64) -> c_int;
    fn cnt_bits_to_n(bits: *const u64, words: usize, len: usize, out: *mut u8) -> c_int;
    fn cnt_n_to_bits2(n: *const u8, n_len: usize, out: *mut u64, out_words: usize) -> c_int;
    fn cnt_n_to_bits2_ex(n: *const u8, n_len: usize, out: *mut u64, out_words: usize, flags: c_uint) -> c_int;
    fn cnt_bits_to_n2(bits: *const u64, words: usize, len: usize, out: *mut u8) -> c_int;
    fn cnt_n_to_bits_sharded(n: *const u8, n_len: usize, out: *mut u64, out_words: usize, ndev: c_int) -> c_int;
    fn cnt_bits_to_n_sharded(bits: *const u64, words: usize, len: usize, out: *mut u8, ndev: c_int) -> c_int;
    fn cnt_n_to_bits2_sharded(n: *const u8, n_len: usize, out: *mut u64, out_words: usize, ndev: c_int) -> c_int;
    fn cnt_bits_to_n2_sharded(bits: *const u64, words: usize, len: usize, out: *mut u8, ndev: c_int) -> c_int;
    // device tier (enqueue-only) + the device-memory helpers a caller without HIP bindings needs
    fn cnt_n_to_bits_dev(d_n: *const c_void, n_len: usize, d_out: *mut c_void, out_words: usize, flags: c_uint, stream: *mut c_void) -> c_int;
    fn cnt_bits_to_n_dev(d_bits: *const c_void, words: usize, len: usize, d_out: *mut c_void, flags: c_uint, stream: *mut c_void) -> c_int;
    fn cnt_n_to_bits_checked_dev(d_n: *const c_void, n_len: usize, d_out: *mut c_void, out_words: usize, flags: c_uint, d_invalid_count: *mut c_void, stream: *mut c_void) -> c_int;
    fn cnt_dev_alloc(d_ptr: *mut *mut c_void, bytes: usize) -> c_int;
    fn cnt_dev_free(d_ptr: *mut c_void) -> c_int;
    fn cnt_dev_upload(d_dst: *mut c_void, h_src: *const c_void, bytes: usize) -> c_int;
    fn cnt_dev_download(h_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    fn cnt_dev_sync(stream: *mut c_void) -> c_int;
    fn cnt_check_device_range(p: *const c_void, bytes: usize, device: c_int) -> c_int;
    fn cnt_set_device(device: c_int) -> c_int;
    fn cnt_shutdown() -> c_int;
    // pinned caller memory: a side of a host-slice call that lies in it is used in place by the copy engines
    fn cnt_host_alloc(p: *mut *mut c_void, bytes: usize) -> c_int;
    fn cnt_host_free(p: *mut c_void) -> c_int;
    fn cnt_host_register(p: *mut c_void, bytes: usize) -> c_int;
    fn cnt_host_unregister(p: *mut c_void) -> c_int;
    fn cnt_host_is_pinned(p: *const c_void, bytes: usize) -> c_int;
    // multi-GPU device tier, enqueue-only: one library stream per shard, any number of ops queued ahead, one wait
    fn cnt_sharded_dev_open(ndev: c_int, flags: c_uint, queue: *mut *mut c_void) -> c_int;
    fn cnt_sharded_dev_open_on_streams(ndev: c_int, streams: *const *mut c_void, flags: c_uint, queue: *mut *mut c_void) -> c_int;
    fn cnt_sharded_dev_open_on_devices(ndev: c_int, devices: *const c_int, flags: c_uint, queue: *mut *mut c_void) -> c_int;
    fn cnt_sharded_dev_device(queue: *mut c_void, k: c_int, device: *mut c_int) -> c_int;
    fn cnt_sharded_dev_shards(queue: *mut c_void, ndev: *mut c_int) -> c_int;
    fn cnt_sharded_dev_wait_event(queue: *mut c_void, k: c_int, event: *mut c_void) -> c_int;
    fn cnt_sharded_dev_record_event(queue: *mut c_void, k: c_int, event: *mut c_void) -> c_int;
    fn cnt_sharded_dev_close(queue: *mut c_void) -> c_int;
    fn cnt_n_to_bits_sharded_dev_enqueue(queue: *mut c_void, d_n: *const *const c_void, n_len: *const usize, d_out: *const *mut c_void, out_words: *const usize, flags: c_uint) -> c_int;
    fn cnt_bits_to_n_sharded_dev_enqueue(queue: *mut c_void, d_bits: *const *const c_void, words: *const usize, len: *const usize, d_out: *const *mut c_void, flags: c_uint) -> c_int;
    fn cnt_n_to_bits_checked_sharded_dev_enqueue(queue: *mut c_void, d_n: *const *const c_void, n_len: *const usize, d_out: *const *mut c_void, out_words: *const usize, flags: c_uint, d_invalid_count: *const *mut c_void) -> c_int;
    fn cnt_sharded_dev_wait(queue: *mut c_void, shard_ms: *mut f32) -> c_int;
    fn cnt_sharded_dev_op_ms(queue: *mut c_void, op: usize, shard_ms: *mut f32) -> c_int;
    // packed-domain operations (host tier): what the reference's README points to, on the packed words
    fn cnt_hamming(a: *const u64, b: *const u64, len: usize, distance: *mut u64) -> c_int;
    fn cnt_complement(bits: *const u64, len: usize, out: *mut u64) -> c_int;
    fn cnt_reverse_complement(bits: *const u64, len: usize, out: *mut u64) -> c_int;
    fn cnt_validate(n: *const u8, n_len: usize, flags: c_uint, invalid: *mut u64) -> c_int;
}

const CNT_STRICT_LUT: c_uint = 1;
const CNT_TAIL_LUT: c_uint = 4;
const CNT_QUEUE_TIMED: c_uint = 1;
/// `*_checked_dev` flag: the counter is `CNT_COUNT_SLOTS` consecutive `u64` (16 KiB, zeroed by the caller) and the count is their
/// sum -- for data in which most 2-KiB tiles hold a stray (one atomic per dirty tile on ONE counter is 32 x the clean time).
pub const CNT_SPREAD_COUNT: c_uint = 8;
pub const CNT_COUNT_SLOTS: usize = 2048;

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

/// `n_to_bits_{pext,shift,movemask,mul}` to the letter, on ANY bytes at any length: bit extraction on whole 32-nt
/// blocks, `BYTE_LUT` on the final partial word (`n_to_bits.rs:109-111,160-162,201-203,253-255`).
pub fn n_to_bits_hip_simd_exact(n: &[u8]) -> Vec<u64> {
    let words = unsafe { cnt_words_for(n.len()) };
    let mut res: Vec<u64> = Vec::with_capacity(words);
    unsafe {
        check(cnt_n_to_bits_ex(n.as_ptr(), n.len(), res.as_mut_ptr(), words, CNT_TAIL_LUT));
        res.set_len(words);
    }
    res
}

/// `n_to_bits_hip` VALIDATED in the same pass over the data: the second value is the number of bytes of `n` outside
/// `ACGTUacgtu` -- what `BYTE_LUT` turns into code 0 without a word (`n_to_bits.rs:8-21,42`; the reference's README points at a
/// separate check, `README.md:23`).  The words are `n_to_bits_hip`'s whatever the count says.
pub fn n_to_bits_hip_checked(n: &[u8]) -> (Vec<u64>, u64) {
    let words = unsafe { cnt_words_for(n.len()) };
    let mut res: Vec<u64> = Vec::with_capacity(words);
    let mut invalid: u64 = 0;
    unsafe {
        check(cnt_n_to_bits_checked(n.as_ptr(), n.len(), res.as_mut_ptr(), words, 0, &mut invalid));
        res.set_len(words);
    }
    (res, invalid)
}

/// The validated encode as a `Result`: `Err(count)` if `n` holds `count > 0` bytes that are not nucleotides.
pub fn try_n_to_bits_hip(n: &[u8]) -> Result<Vec<u64>, u64> {
    let (res, invalid) = n_to_bits_hip_checked(n);
    if invalid == 0 {
        Ok(res)
    } else {
        Err(invalid)
    }
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

/// `n_to_bits2_hip` validated in the same pass: the second value counts the bytes outside `ACGTUNacgtun` (`n_to_bits2.rs:8-23`).
pub fn n_to_bits2_hip_checked(n: &[u8]) -> (Vec<u64>, u64) {
    let words = unsafe { cnt_words2_for(n.len()) };
    let mut res: Vec<u64> = Vec::with_capacity(words);
    let mut invalid: u64 = 0;
    unsafe {
        check(cnt_n_to_bits2_checked(n.as_ptr(), n.len(), res.as_mut_ptr(), words, 0, &mut invalid));
        res.set_len(words);
    }
    (res, invalid)
}

/// `n_to_bits2_pext` to the letter on any bytes: its low-3-bit table up to word `(len - 5) / 27`, `BYTE_LUT` from there
/// on (`n_to_bits2.rs:120,179-185`).
pub fn n_to_bits2_hip_pext_exact(n: &[u8]) -> Vec<u64> {
    let words = unsafe { cnt_words2_for(n.len()) };
    let mut res: Vec<u64> = Vec::with_capacity(words);
    unsafe {
        check(cnt_n_to_bits2_ex(n.as_ptr(), n.len(), res.as_mut_ptr(), words, CNT_TAIL_LUT));
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

/// `_into` forms: the same four codecs writing into a `Vec` the CALLER owns.  The `Vec` is cleared and its capacity
/// reused (grown without zero-fill when too small), so a loop that times like the reference's harness does -- the result
/// allocated AND dropped inside the timed call (`benches/bench_n_to_bits.rs:6-7`) -- pays neither the page faults of a
/// fresh allocation nor the `munmap` of the dropped one: at 1 GiB that is 22 ms per decode instead of 74 ms
/// (BENCH_r03 `host_tier`).  Same results, same panics as the returning forms.
pub fn n_to_bits_hip_into(n: &[u8], res: &mut Vec<u64>) {
    let words = unsafe { cnt_words_for(n.len()) };
    res.clear();
    res.reserve(words);
    unsafe {
        check(cnt_n_to_bits_ex(n.as_ptr(), n.len(), res.as_mut_ptr(), words, 0));
        res.set_len(words);
    }
}

pub fn bits_to_n_hip_into(bits: &[u64], len: usize, res: &mut Vec<u8>) {
    if len > (bits.len() << 5) {
        panic!("The length is greater than the number of nucleotides!");
    }
    res.clear();
    res.reserve(len);
    unsafe {
        check(cnt_bits_to_n(bits.as_ptr(), bits.len(), len, res.as_mut_ptr()));
        res.set_len(len);
    }
}

pub fn n_to_bits2_hip_into(n: &[u8], res: &mut Vec<u64>) {
    let words = unsafe { cnt_words2_for(n.len()) };
    res.clear();
    res.reserve(words);
    unsafe {
        check(cnt_n_to_bits2(n.as_ptr(), n.len(), res.as_mut_ptr(), words));
        res.set_len(words);
    }
}

pub fn bits_to_n2_hip_into(bits: &[u64], len: usize, res: &mut Vec<u8>) {
    if len > bits.len() * 27 {
        panic!("The length is greater than the number of nucleotides!");
    }
    res.clear();
    res.reserve(len);
    unsafe {
        check(cnt_bits_to_n2(bits.as_ptr(), bits.len(), len, res.as_mut_ptr()));
        res.set_len(len);
    }
}

/// Slice forms: the 2-bit codec between slices the CALLER owns -- the forms that can use PINNED memory (`PinnedBuf`, `HostPin`):
/// a side that lies in pinned memory is not staged, the copy engines read / write it in place (include/cute_nt.h "pinned caller
/// memory"; 4-10 % at 2^28 nt and above, profiles/r06_host_tier.md).  `n_to_bits_hip_slice` returns the number of words written
/// (`out` must hold `words_for(n.len())`), `bits_to_n_hip_slice` writes `len` bytes.  Same results, same panics.
pub fn n_to_bits_hip_slice(n: &[u8], out: &mut [u64]) -> usize {
    let words = unsafe { cnt_words_for(n.len()) };
    assert!(out.len() >= words);
    unsafe { check(cnt_n_to_bits_ex(n.as_ptr(), n.len(), out.as_mut_ptr(), out.len(), 0)) };
    words
}

pub fn bits_to_n_hip_slice(bits: &[u64], len: usize, out: &mut [u8]) {
    if len > (bits.len() << 5) {
        panic!("The length is greater than the number of nucleotides!");
    }
    assert!(out.len() >= len);
    unsafe { check(cnt_bits_to_n(bits.as_ptr(), bits.len(), len, out.as_mut_ptr())) };
}

/// `len` elements of pinned, device-mapped host memory (`cnt_host_alloc`), zero-filled; derefs to a slice.  For buffers that
/// live as long as the pipeline: pinned memory cannot be swapped.
pub struct PinnedBuf<T: Copy> {
    ptr: *mut T,
    len: usize,
}

impl<T: Copy> PinnedBuf<T> {
    pub fn new(len: usize) -> PinnedBuf<T> {
        let bytes = len.checked_mul(std::mem::size_of::<T>()).expect("PinnedBuf: size overflow");
        assert!(bytes > 0);
        let mut p: *mut c_void = std::ptr::null_mut();
        unsafe {
            check(cnt_host_alloc(&mut p, bytes));
            std::ptr::write_bytes(p as *mut u8, 0, bytes);
        }
        PinnedBuf { ptr: p as *mut T, len }
    }
}

impl<T: Copy> std::ops::Deref for PinnedBuf<T> {
    type Target = [T];
    fn deref(&self) -> &[T] {
        unsafe { std::slice::from_raw_parts(self.ptr, self.len) }
    }
}

impl<T: Copy> std::ops::DerefMut for PinnedBuf<T> {
    fn deref_mut(&mut self) -> &mut [T] {
        unsafe { std::slice::from_raw_parts_mut(self.ptr, self.len) }
    }
}

impl<T: Copy> Drop for PinnedBuf<T> {
    fn drop(&mut self) {
        unsafe { cnt_host_free(self.ptr as *mut c_void) };
    }
}

/// Pins an EXISTING slice in place for the guard's lifetime (`cnt_host_register`: tens of milliseconds per GiB, once;
/// unregistered on drop).  The borrow keeps the memory from being freed or moved while it is pinned.
pub struct HostPin<'a, T> {
    slice: &'a mut [T],
}

impl<'a, T> HostPin<'a, T> {
    pub fn new(slice: &'a mut [T]) -> HostPin<'a, T> {
        assert!(!slice.is_empty());
        unsafe { check(cnt_host_register(slice.as_mut_ptr() as *mut c_void, std::mem::size_of_val(slice))) };
        HostPin { slice }
    }

    pub fn get(&mut self) -> &mut [T] {
        self.slice
    }
}

impl<'a, T> Drop for HostPin<'a, T> {
    fn drop(&mut self) {
        unsafe { cnt_host_unregister(self.slice.as_mut_ptr() as *mut c_void) };
    }
}

/// Would the host tier use this slice in place?
pub fn is_pinned<T>(s: &[T]) -> bool {
    unsafe { cnt_host_is_pinned(s.as_ptr() as *const c_void, std::mem::size_of_val(s)) == 1 }
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

// ---- operations on the packed words without decoding (README.md:20-25,45 of the reference) ----------------

fn need(bits: &[u64], len: usize) {
    if len > (bits.len() << 5) {
        panic!("The length is greater than the number of nucleotides!");
    }
}

/// Number of positions `i < len` whose 2-bit codes differ.
pub fn hamming_hip(a: &[u64], b: &[u64], len: usize) -> u64 {
    need(a, len);
    need(b, len);
    let mut d: u64 = 0;
    unsafe { check(cnt_hamming(a.as_ptr(), b.as_ptr(), len, &mut d)) };
    d
}

/// A<->T, C<->G on the packed words; unused high bits of the last word stay zero.
pub fn complement_hip(bits: &[u64], len: usize) -> Vec<u64> {
    need(bits, len);
    let words = unsafe { cnt_words_for(len) };
    let mut res: Vec<u64> = Vec::with_capacity(words);
    unsafe {
        check(cnt_complement(bits.as_ptr(), len, res.as_mut_ptr()));
        res.set_len(words);
    }
    res
}

/// `out(i) = complement(in(len - 1 - i))`.
pub fn reverse_complement_hip(bits: &[u64], len: usize) -> Vec<u64> {
    need(bits, len);
    let words = unsafe { cnt_words_for(len) };
    let mut res: Vec<u64> = Vec::with_capacity(words);
    unsafe {
        check(cnt_reverse_complement(bits.as_ptr(), len, res.as_mut_ptr()));
        res.set_len(words);
    }
    res
}

/// Number of bytes outside `ACGTUacgtu` (with `allow_n` also `N`/`n` are legal); 0 = a valid sequence.
pub fn validate_hip(n: &[u8], allow_n: bool) -> u64 {
    let mut bad: u64 = 0;
    unsafe { check(cnt_validate(n.as_ptr(), n.len(), if allow_n { 2 } else { 0 }, &mut bad)) };
    bad
}

// ---- device-resident tier: data already in HBM (what the roofline numbers measure) ------------------

/// Number of packed words for `n_len` nucleotides (`ceil(n_len / 32)`, n_to_bits.rs:35).
pub fn words_for(n_len: usize) -> usize {
    unsafe { cnt_words_for(n_len) }
}

/// Owned device memory on the calling thread's current device (hipMalloc / hipFree behind the C ABI).
pub struct DeviceBuffer {
    ptr: *mut c_void,
    bytes: usize,
}

impl DeviceBuffer {
    pub fn new(bytes: usize) -> DeviceBuffer {
        let mut ptr: *mut c_void = std::ptr::null_mut();
        unsafe { check(cnt_dev_alloc(&mut ptr, bytes)) };
        DeviceBuffer { ptr, bytes }
    }

    pub fn from_slice<T: Copy>(host: &[T]) -> DeviceBuffer {
        let bytes = std::mem::size_of_val(host);
        let buf = DeviceBuffer::new(bytes);
        unsafe { check(cnt_dev_upload(buf.ptr, host.as_ptr() as *const c_void, bytes)) };
        buf
    }

    /// Copies the first `len` elements back to the host (synchronous).
    pub fn to_vec<T: Copy>(&self, len: usize) -> Vec<T> {
        let bytes = len * std::mem::size_of::<T>();
        assert!(bytes <= self.bytes);
        let mut res: Vec<T> = Vec::with_capacity(len);
        unsafe {
            check(cnt_dev_download(res.as_mut_ptr() as *mut c_void, self.ptr, bytes));
            res.set_len(len);
        }
        res
    }
}

impl Drop for DeviceBuffer {
    fn drop(&mut self) {
        unsafe { cnt_dev_free(self.ptr) };
    }
}

/// Enqueue the encode of `n_len` device-resident nucleotides into `d_out` (>= `words_for(n_len)` words) on
/// the default stream; returns without waiting (`device_sync()` waits).
pub fn n_to_bits_hip_dev(d_n: &DeviceBuffer, n_len: usize, d_out: &DeviceBuffer) {
    assert!(n_len <= d_n.bytes);
    unsafe { check(cnt_n_to_bits_dev(d_n.ptr, n_len, d_out.ptr, d_out.bytes / 8, 0, std::ptr::null_mut())) };
}

/// Encode + validity count in ONE pass over the resident ASCII: the launch ADDS the number of bytes outside `ACGTUacgtu`
/// to the `u64` at the start of `d_invalid` (>= 8 bytes of device memory, zeroed by the caller).
pub fn n_to_bits_hip_checked_dev(d_n: &DeviceBuffer, n_len: usize, d_out: &DeviceBuffer, d_invalid: &DeviceBuffer) {
    assert!(n_len <= d_n.bytes && d_invalid.bytes >= 8);
    unsafe { check(cnt_n_to_bits_checked_dev(d_n.ptr, n_len, d_out.ptr, d_out.bytes / 8, 0, d_invalid.ptr, std::ptr::null_mut())) };
}

/// The same with the count spread over `CNT_COUNT_SLOTS` counters (`d_invalid`: >= 16 KiB, zeroed by the caller; the count is the
/// sum of its first `CNT_COUNT_SLOTS` words): what to use when the data is expected to be dirty.
pub fn n_to_bits_hip_checked_dev_spread(d_n: &DeviceBuffer, n_len: usize, d_out: &DeviceBuffer, d_invalid: &DeviceBuffer) {
    assert!(n_len <= d_n.bytes && d_invalid.bytes >= 8 * CNT_COUNT_SLOTS);
    unsafe { check(cnt_n_to_bits_checked_dev(d_n.ptr, n_len, d_out.ptr, d_out.bytes / 8, CNT_SPREAD_COUNT, d_invalid.ptr, std::ptr::null_mut())) };
}

/// Enqueue the decode of `len` nucleotides from `words` device-resident words into `d_out` (>= `len` bytes).
pub fn bits_to_n_hip_dev(d_bits: &DeviceBuffer, words: usize, len: usize, d_out: &DeviceBuffer) {
    if len > (words << 5) {
        panic!("The length is greater than the number of nucleotides!");
    }
    assert!(words * 8 <= d_bits.bytes && len <= d_out.bytes);
    unsafe { check(cnt_bits_to_n_dev(d_bits.ptr, words, len, d_out.ptr, 0, std::ptr::null_mut())) };
}

/// Make `device` the calling thread's current device (what `DeviceBuffer::new` allocates on).
pub fn set_device(device: i32) {
    unsafe { check(cnt_set_device(device)) };
}

/// Enqueue-only multi-GPU device tier: shard `k` is resident on device `k`; every `enqueue_*` call queues that codec
/// call on every shard's own library stream and returns at once, `wait` drains all of them.  Queue the decode of step
/// `s` behind its encode and step `s + 1` behind that, wait once: the devices run back to back while the host is a
/// whole queue ahead (word `w` depends on nucleotides `[32w, 32w + 32)` only, `n_to_bits.rs:38-43`: no barrier).
pub struct ShardedDevQueue {
    handle: *mut c_void,
    ndev: usize,
}

impl ShardedDevQueue {
    /// `ndev` shards on devices `0..ndev` (`0` = all visible devices); `timed` records an event behind every op
    /// (`wait` / `op_ms` report device ms; at most 4096 ops per batch).  `self.ndev` is the shard count the LIBRARY
    /// resolved (`cnt_sharded_dev_shards`), never the caller's `0`: `wait` / `op_ms` size their buffers by it.
    pub fn new(ndev: usize, timed: bool) -> ShardedDevQueue {
        assert!(ndev <= c_int::MAX as usize);
        let mut handle: *mut c_void = std::ptr::null_mut();
        unsafe { check(cnt_sharded_dev_open(ndev as c_int, if timed { CNT_QUEUE_TIMED } else { 0 }, &mut handle)) };
        ShardedDevQueue::adopt(handle)
    }

    /// The queue adopts the caller's streams (`streams[k]`: a non-null `hipStream_t` of device `k`): shard `k`'s ops run in
    /// order with whatever else the caller enqueues there -- no host synchronisation between producer, codec and consumer.
    pub fn on_streams(streams: &[*mut c_void], timed: bool) -> ShardedDevQueue {
        assert!(!streams.is_empty() && streams.len() <= c_int::MAX as usize && streams.iter().all(|s| !s.is_null()));
        let mut handle: *mut c_void = std::ptr::null_mut();
        unsafe { check(cnt_sharded_dev_open_on_streams(streams.len() as c_int, streams.as_ptr(), if timed { CNT_QUEUE_TIMED } else { 0 }, &mut handle)) };
        ShardedDevQueue::adopt(handle)
    }

    /// Library-owned streams on an explicit device list: shard `k` runs on `devices[k]` (any subset, order or repetition of
    /// the visible devices).
    pub fn on_devices(devices: &[i32], timed: bool) -> ShardedDevQueue {
        assert!(!devices.is_empty() && devices.len() <= c_int::MAX as usize);
        let mut handle: *mut c_void = std::ptr::null_mut();
        unsafe { check(cnt_sharded_dev_open_on_devices(devices.len() as c_int, devices.as_ptr(), if timed { CNT_QUEUE_TIMED } else { 0 }, &mut handle)) };
        ShardedDevQueue::adopt(handle)
    }

    /// The device the queue runs shard `k` on -- for adopted streams what the STREAM says, not `k`.
    pub fn device(&self, k: usize) -> i32 {
        assert!(k < self.ndev);
        let mut d: c_int = -1;
        unsafe { check(cnt_sharded_dev_device(self.handle, k as c_int, &mut d)) };
        d
    }

    fn adopt(handle: *mut c_void) -> ShardedDevQueue {
        let mut n: c_int = 0;
        unsafe { check(cnt_sharded_dev_shards(handle, &mut n)) };
        assert!(n >= 1);
        ShardedDevQueue { handle, ndev: n as usize }
    }

    /// Number of shards this queue drives.
    pub fn shards(&self) -> usize {
        self.ndev
    }

    /// Shard `k`'s stream waits ON THE DEVICE for `event` (a `hipEvent_t` the caller recorded behind the work that
    /// produces shard `k`'s buffer); ops enqueued afterwards run behind it.  The host does not wait.
    pub fn wait_event(&mut self, k: usize, event: *mut c_void) {
        assert!(k < self.ndev && !event.is_null());
        unsafe { check(cnt_sharded_dev_wait_event(self.handle, k as c_int, event)) };
    }

    /// Record the caller's `hipEvent_t` (of shard `k`'s device) behind everything queued for shard `k` so far.
    pub fn record_event(&mut self, k: usize, event: *mut c_void) {
        assert!(k < self.ndev && !event.is_null());
        unsafe { check(cnt_sharded_dev_record_event(self.handle, k as c_int, event)) };
    }

    /// Queue the encode of every shard: `d_n[k]` holds `n_len[k]` nucleotides on device `k`, `d_out[k]` its words.
    pub fn enqueue_n_to_bits(&mut self, d_n: &[&DeviceBuffer], n_len: &[usize], d_out: &[&DeviceBuffer]) {
        assert!(d_n.len() == self.ndev && n_len.len() == self.ndev && d_out.len() == self.ndev);
        let ins: Vec<*const c_void> = d_n.iter().map(|b| b.ptr as *const c_void).collect();
        let outs: Vec<*mut c_void> = d_out.iter().map(|b| b.ptr).collect();
        let caps: Vec<usize> = d_out.iter().map(|b| b.bytes / 8).collect();
        for k in 0..self.ndev {
            assert!(n_len[k] <= d_n[k].bytes);
        }
        unsafe { check(cnt_n_to_bits_sharded_dev_enqueue(self.handle, ins.as_ptr(), n_len.as_ptr(), outs.as_ptr(), caps.as_ptr(), 0)) };
    }

    /// Queue the VALIDATED encode of every shard: as `enqueue_n_to_bits`, and shard `k`'s op adds the number of its bytes
    /// outside `ACGTUacgtu` to the `u64` in `d_invalid[k]` (>= 8 bytes on shard `k`'s device, zeroed by the caller).
    pub fn enqueue_n_to_bits_checked(&mut self, d_n: &[&DeviceBuffer], n_len: &[usize], d_out: &[&DeviceBuffer], d_invalid: &[&DeviceBuffer]) {
        assert!(d_n.len() == self.ndev && n_len.len() == self.ndev && d_out.len() == self.ndev && d_invalid.len() == self.ndev);
        let ins: Vec<*const c_void> = d_n.iter().map(|b| b.ptr as *const c_void).collect();
        let outs: Vec<*mut c_void> = d_out.iter().map(|b| b.ptr).collect();
        let caps: Vec<usize> = d_out.iter().map(|b| b.bytes / 8).collect();
        let counters: Vec<*mut c_void> = d_invalid.iter().map(|b| b.ptr).collect();
        for k in 0..self.ndev {
            assert!(n_len[k] <= d_n[k].bytes && d_invalid[k].bytes >= 8);
        }
        unsafe { check(cnt_n_to_bits_checked_sharded_dev_enqueue(self.handle, ins.as_ptr(), n_len.as_ptr(), outs.as_ptr(), caps.as_ptr(), 0, counters.as_ptr())) };
    }

    /// Queue the decode of every shard: `len[k]` nucleotides from `words[k]` words of `d_bits[k]` into `d_out[k]`.
    pub fn enqueue_bits_to_n(&mut self, d_bits: &[&DeviceBuffer], words: &[usize], len: &[usize], d_out: &[&DeviceBuffer]) {
        assert!(d_bits.len() == self.ndev && words.len() == self.ndev && len.len() == self.ndev && d_out.len() == self.ndev);
        for k in 0..self.ndev {
            if len[k] > (words[k] << 5) {
                panic!("The length is greater than the number of nucleotides!");
            }
            assert!(words[k] * 8 <= d_bits[k].bytes && len[k] <= d_out[k].bytes);
        }
        let ins: Vec<*const c_void> = d_bits.iter().map(|b| b.ptr as *const c_void).collect();
        let outs: Vec<*mut c_void> = d_out.iter().map(|b| b.ptr).collect();
        unsafe { check(cnt_bits_to_n_sharded_dev_enqueue(self.handle, ins.as_ptr(), words.as_ptr(), len.as_ptr(), outs.as_ptr(), 0)) };
    }

    /// Wait for everything queued; per-shard device milliseconds of the batch (zeros unless `timed`).
    pub fn wait(&mut self) -> Vec<f32> {
        let mut ms = vec![0f32; self.ndev];
        unsafe { check(cnt_sharded_dev_wait(self.handle, ms.as_mut_ptr())) };
        ms
    }

    /// Per-shard device milliseconds of op `op` of the batch the last `wait` drained (`timed` queues).
    pub fn op_ms(&self, op: usize) -> Vec<f32> {
        let mut ms = vec![0f32; self.ndev];
        unsafe { check(cnt_sharded_dev_op_ms(self.handle, op, ms.as_mut_ptr())) };
        ms
    }
}

impl Drop for ShardedDevQueue {
    fn drop(&mut self) {
        unsafe { cnt_sharded_dev_close(self.handle) };
    }
}

/// Debug aid for code that hands RAW device pointers to the `*_dev` entry points (a host pointer there is a GPU page
/// fault when the kernel runs, not an error): true if `[p, p + bytes)` lies inside one allocation the current device
/// can address.
pub fn check_device_range(p: *const c_void, bytes: usize) -> bool {
    unsafe { cnt_check_device_range(p, bytes, -1) == 0 }
}

/// Wait for everything enqueued on the default stream.
pub fn device_sync() {
    unsafe { check(cnt_dev_sync(std::ptr::null_mut())) };
}

/// Release the calling thread's cached streams / staging buffers (and the sharded tier's workers').
pub fn shutdown() {
    unsafe { check(cnt_shutdown()) };
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
    fn test_packed_ops() {
        let x = n_to_bits_hip(b"ATCGATCG");
        assert_eq!(hamming_hip(&x, &x, 8), 0);
        assert_eq!(hamming_hip(&x, &complement_hip(&x, 8), 8), 8);
        assert_eq!(bits_to_n_hip(&complement_hip(&x, 8), 8), b"TAGCTAGC".to_vec());
        assert_eq!(bits_to_n_hip(&reverse_complement_hip(&x, 8), 8), b"CGATCGAT".to_vec());
        assert_eq!(validate_hip(b"ACGTN", false), 1);
        assert_eq!(validate_hip(b"ACGTN", true), 0);
    }

    #[test]
    fn test_device_resident_round_trip() {
        let n = b"ATCGATCGATCGATCGATCGATCGATCGATCG";
        let d_n = DeviceBuffer::from_slice(&n[..]);
        let d_bits = DeviceBuffer::new(8);
        let d_back = DeviceBuffer::new(32);
        n_to_bits_hip_dev(&d_n, 32, &d_bits);
        bits_to_n_hip_dev(&d_bits, 1, 32, &d_back);
        device_sync();
        assert_eq!(d_bits.to_vec::<u64>(1), vec![0xD8D8D8D8D8D8D8D8u64]);
        assert_eq!(d_back.to_vec::<u8>(32), n.to_vec());
    }

    #[test]
    fn test_pinned_slices_give_the_same_words() {
        let mut n: PinnedBuf<u8> = PinnedBuf::new(1 << 22);
        for (i, b) in n.iter_mut().enumerate() {
            *b = b"ATCG"[i & 3];
        }
        assert!(is_pinned(&n[..]) && !is_pinned(&vec![0u8; 4096][..]));
        let mut out: PinnedBuf<u64> = PinnedBuf::new(words_for(n.len()));
        let words = n_to_bits_hip_slice(&n, &mut out);
        assert_eq!(words, 1 << 17);
        assert!(out.iter().all(|&w| w == 0xD8D8D8D8D8D8D8D8));
        let mut back = vec![0u8; n.len()];
        {
            let mut pin = HostPin::new(&mut back[..]);
            bits_to_n_hip_slice(&out, 1 << 22, pin.get());
        }
        assert_eq!(&back[..], &n[..]);
    }

    #[test]
    fn test_into_forms_reuse_the_callers_vec() {
        let mut words: Vec<u64> = Vec::new();
        n_to_bits_hip_into(b"ATCGATCGATCGATCGATCGATCGATCGATCG", &mut words);
        assert_eq!(words, vec![0b1101100011011000110110001101100011011000110110001101100011011000]);
        let cap = words.capacity();
        n_to_bits_hip_into(b"ATCG", &mut words);
        assert_eq!(words, vec![0b11011000]);
        assert_eq!(words.capacity(), cap);
        let mut n: Vec<u8> = Vec::new();
        bits_to_n_hip_into(&vec![0b1101100011011000110110001101100011011000110110001101100011011000], 32, &mut n);
        assert_eq!(n, b"ATCGATCGATCGATCGATCGATCGATCGATCG");
        let mut w2: Vec<u64> = Vec::new();
        n_to_bits2_hip_into(b"ATCGN", &mut w2);
        assert_eq!(w2, vec![0b101110100011]);
        bits_to_n2_hip_into(&w2, 5, &mut n);
        assert_eq!(n, b"ATCGN");
    }

    #[test]
    fn test_sharded_dev_queue_steps_ahead() {
        set_device(0);
        let n = b"ATCGATCGATCGATCGATCGATCGATCGATCG".repeat(1 << 10);
        let d_n = DeviceBuffer::from_slice(&n);
        let d_bits = DeviceBuffer::new(n.len() / 4);
        let d_back = DeviceBuffer::new(n.len());
        let mut q = ShardedDevQueue::new(1, true);
        for _ in 0..3 {
            q.enqueue_n_to_bits(&[&d_n], &[n.len()], &[&d_bits]);
            q.enqueue_bits_to_n(&[&d_bits], &[n.len() / 32], &[n.len()], &[&d_back]);
        }
        let ms = q.wait();
        assert!(ms[0] > 0.0 && q.op_ms(5)[0] > 0.0);
        assert_eq!(d_back.to_vec::<u8>(n.len()), n);
        assert!(d_bits.to_vec::<u64>(n.len() / 32).iter().all(|w| *w == 0xD8D8D8D8D8D8D8D8));
    }

    #[test]
    fn test_checked_encode_counts_what_byte_lut_zeroes() {
        assert_eq!(n_to_bits_hip_checked(b"ATCGATCG"), (vec![0xD8D8u64], 0));
        let (words, invalid) = n_to_bits_hip_checked(b"ATCGNNxx");
        assert_eq!(invalid, 4);
        assert_eq!(words, n_to_bits_hip(b"ATCGNNxx"));
        assert_eq!(try_n_to_bits_hip(b"ATCGN"), Err(1));
        assert_eq!(try_n_to_bits_hip(b"acgu"), Ok(n_to_bits_hip(b"ACGT")));
        assert_eq!(n_to_bits2_hip_checked(b"ATCGN").1, 0);
        assert_eq!(n_to_bits2_hip_checked(b"ATCGNx-").1, 2);
        let d_n = DeviceBuffer::from_slice(&b"ATCGATCGATCGATCGATCGATCGATCGAT?G"[..]);
        let d_bits = DeviceBuffer::new(8);
        let d_bad = DeviceBuffer::from_slice(&[0u64][..]);
        n_to_bits_hip_checked_dev(&d_n, 32, &d_bits, &d_bad);
        device_sync();
        assert_eq!(d_bad.to_vec::<u64>(1), vec![1u64]);
    }

    #[test]
    fn test_queue_on_an_explicit_device_list() {
        let n = b"ATCGATCGATCGATCGATCGATCGATCGATCN".repeat(64);
        let mut q = ShardedDevQueue::on_devices(&[0, 0], false);
        assert_eq!((q.shards(), q.device(0), q.device(1)), (2, 0, 0));
        set_device(0);
        let d_n = DeviceBuffer::from_slice(&n);
        let (o0, o1) = (DeviceBuffer::new(n.len() / 4), DeviceBuffer::new(n.len() / 4));
        let (c0, c1) = (DeviceBuffer::from_slice(&[0u64][..]), DeviceBuffer::from_slice(&[0u64][..]));
        q.enqueue_n_to_bits_checked(&[&d_n, &d_n], &[n.len(), n.len() - 32], &[&o0, &o1], &[&c0, &c1]);
        q.wait();
        assert_eq!((c0.to_vec::<u64>(1)[0], c1.to_vec::<u64>(1)[0]), (64, 63));
    }

    #[test]
    fn test_bits_to_n2_hip() {
        assert_eq!(bits_to_n2_hip(&vec![0x36A45D1F46D48BA3u64, 0x5D1F4], 35), "ATCGNATCGNATCGNATCGNATCGNATCGNATCGN".as_bytes());
    }
}
