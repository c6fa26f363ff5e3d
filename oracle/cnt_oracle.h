/*
 * cnt_oracle.h -- CPU oracle for the nucleotide codec hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The shipped path
 * (cute_nucleotides_amd/, libcute_nt_hip.so) never links or calls it.
 *
 * What it is: a plain-C restatement of the reference's algorithms
 * (/root/reference/src/n_to_bits.rs, /root/reference/src/n_to_bits2.rs), one C
 * function per reference function, each citing the lines it follows.  The
 * reference is Rust; no Rust toolchain exists in this image, so the reference
 * itself cannot be built here (oracle/_ref is therefore absent -- see
 * oracle/README.md).  Parity is pinned instead by every known-answer vector the
 * reference's own unit tests hold (tests/golden/reference_kats.json).
 *
 * Calling convention: caller-allocated outputs, plain pointers + sizes.
 * Return 0 on success, CNT_ORACLE_ELEN when a decoder's `len` exceeds capacity
 * (the reference panics: n_to_bits.rs:52-54), CNT_ORACLE_ECAP when the output
 * buffer is too small.
 */
#ifndef CNT_ORACLE_H
#define CNT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNT_ORACLE_OK 0
#define CNT_ORACLE_ELEN 1 /* "The length is greater than the number of nucleotides!" */
#define CNT_ORACLE_ECAP 2 /* caller's output buffer too small */
#define CNT_ORACLE_ECPU 3 /* SIMD port called on a CPU without AVX2/BMI2/PCLMUL */

/* ceil(n_len/32): words produced by every 2-bit encoder (n_to_bits.rs:35,83). */
size_t cnt_oracle_words_for(size_t n_len);
/* ceil(n_len/27): words produced by the 5-letter encoders (n_to_bits2.rs:38,120). */
size_t cnt_oracle_words2_for(size_t n_len);

/* ---- 2-bit codec, scalar (THE parity oracle) ------------------------------ */
/* n_to_bits.rs:8-21,34-47.  Bytes outside {ACGTUacgtu} encode as 0; bytes
 * >= 0x80 (out-of-bounds LUT read = UB in the reference) are DEFINED as 0. */
int cnt_oracle_n_to_bits_lut(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words);
/* n_to_bits.rs:23-30,51-69.  Writes exactly `len` bytes. */
int cnt_oracle_bits_to_n_lut(const uint64_t *bits, size_t words, size_t len, uint8_t *out);

/* The function every SIMD encoder of the reference computes (mask 0x06 at
 * n_to_bits.rs:85,130,222; shifts 6/5 + movemask at :187-196): code =
 * (byte >> 1) & 3 for EVERY byte value, tail through the LUT (:109-111).
 * Scalar, no intrinsics; used to pin the "fast" HIP encode mode on arbitrary
 * bytes.  Identical to the LUT on the valid alphabet. */
int cnt_oracle_n_to_bits_bitextract(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words);

/* ---- 5-letter codec, scalar ---------------------------------------------- */
/* n_to_bits2.rs:8-23,37-74.  3 nt -> a + 5b + 25c (7 bits); 9 triplets/word. */
int cnt_oracle_n_to_bits2_lut(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words);
/* n_to_bits2.rs:25-33,78-107.  Writes exactly `len` bytes. */
int cnt_oracle_bits_to_n2_lut(const uint64_t *bits, size_t words, size_t len, uint8_t *out);

/* ---- x86 SIMD ports (timed CPU baseline; cnt_simd_port.c) ------------------ */
/* Same intrinsics as the reference (Rust std::arch::x86_64 == <immintrin.h>).
 * Encoders: out_words >= ceil(n_len/32).  Decoders: `out` must hold words*32
 * bytes (the reference always stores whole 32-byte blocks, n_to_bits.rs:271,298)
 * and be 32-byte aligned. */
int cnt_port_cpu_ok(void);
int cnt_port_n_to_bits_pext(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words);     /* :80-115  */
int cnt_port_n_to_bits_shift(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words);    /* :121-166 */
int cnt_port_n_to_bits_movemask(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words); /* :172-207 */
int cnt_port_n_to_bits_mul(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words);      /* :213-259 */
int cnt_port_bits_to_n_shuffle(const uint64_t *bits, size_t words, size_t len, uint8_t *out);     /* :265-303 */
int cnt_port_bits_to_n_pdep(const uint64_t *bits, size_t words, size_t len, uint8_t *out);        /* :309-340 */
int cnt_port_bits_to_n_clmul(const uint64_t *bits, size_t words, size_t len, uint8_t *out);       /* :346-404 */
/* n_to_bits2.rs:118-189 / :196-268.  Decoder `out` must hold words*27+5 bytes. */
int cnt_port_n_to_bits2_pext(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words);
int cnt_port_bits_to_n2_pdep(const uint64_t *bits, size_t words, size_t len, uint8_t *out);

/* ---- packed-domain operations (NOT in the reference: parity unpinned) ----
 * Scalar definitions, one nucleotide at a time; see include/cute_nt.h for the semantics. */
uint64_t cnt_oracle_hamming(const uint64_t *a, const uint64_t *b, size_t len);
void cnt_oracle_complement(const uint64_t *bits, size_t len, uint64_t *out);
void cnt_oracle_reverse_complement(const uint64_t *bits, size_t len, uint64_t *out);
uint64_t cnt_oracle_validate(const uint8_t *n, size_t n_len, int allow_n);

/* Reference-faithful timing: seconds per call with the output malloc'ed and freed
 * INSIDE the timed call, as benches/bench_n_to_bits.rs:6-7 demands.
 * fn: 0 lut 1 pext 2 shift 3 movemask 4 mul 5 memcpy (in = ASCII, n_len nt)
 *     10 lut 11 shuffle 12 pdep 13 clmul       (in = packed words, n_len nt) */
double cnt_port_time_alloc_inclusive(int fn, const void *in, size_t n_len, int iters);

/* ---- shared synthetic-input generator + checksum --------------------------- */
/* Counter-based uniform {A,C,G,T}: block w (32 nt) draws r = splitmix64(seed +
 * w*0x9E3779B97F4A7C15) and emits "ACGT"[(r >> 2k) & 3] for k = 0..31.  The HIP
 * library has the same generator on device (cnt_fill_random_acgt_dev), so a host
 * can regenerate any chunk of a device-resident buffer without a PCIe copy.
 * `first_nt` must be a multiple of 32. */
void cnt_oracle_fill_random_acgt(uint8_t *out, size_t first_nt, size_t n_len, uint64_t seed);
/* Same, over {A,C,G,T,N} (5-letter workloads): 27-nt block w draws the same
 * splitmix64 stream and takes nt k from (r >> 2k)&3, replaced by 'N' when
 * bits [54+..] say so -- see the .c file. `first_nt` multiple of 27. */
void cnt_oracle_fill_random_acgtn(uint8_t *out, size_t first_nt, size_t n_len, uint64_t seed);

/* Position-salted 64-bit checksum over u64 words: sum_i mix(w[i] ^ mix0(first_word+i))
 * mod 2^64 -- order-independent (parallel-friendly), position-sensitive. */
uint64_t cnt_oracle_checksum_words(const uint64_t *w, size_t first_word, size_t words);

#ifdef __cplusplus
}
#endif
#endif /* CNT_ORACLE_H */
