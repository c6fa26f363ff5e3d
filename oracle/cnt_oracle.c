/*
 * cnt_oracle.c -- scalar CPU restatement of the reference codec (TEST
 * INFRASTRUCTURE ONLY; see cnt_oracle.h).  Plain C11, no intrinsics.
 *
 * Each function follows the cited lines of /root/reference/src/n_to_bits.rs or
 * /root/reference/src/n_to_bits2.rs.  Parity status: PINNED by the reference's
 * own known-answer tests (tests/golden/reference_kats.json, checked by
 * tests/test_oracle.py).
 */
#include "cnt_oracle.h"

#include <string.h>

/* ---- lookup tables -------------------------------------------------------- */

/* n_to_bits.rs:8-21 -- a,A=00  c,C=01  t,T,u,U=10  g,G=11, every other 7-bit
 * byte 0.  The reference table has 128 entries and is indexed unchecked by a
 * raw byte (:42), i.e. UB for bytes >= 0x80; here those are defined as 0. */
static const uint8_t BYTE_LUT[256] = {
    ['a'] = 0, ['A'] = 0, ['c'] = 1, ['C'] = 1, ['t'] = 2, ['T'] = 2, ['u'] = 2, ['U'] = 2, ['g'] = 3, ['G'] = 3,
};
static inline uint8_t byte_lut(uint8_t c) { return BYTE_LUT[c]; }

/* n_to_bits.rs:23-30 -- 00=A 01=C 10=T 11=G. */
static const uint8_t BITS_LUT[4] = {'A', 'C', 'T', 'G'};

/* n_to_bits2.rs:8-23 -- A0 C1 T/U2 G3 N4 (note: NOT the 2-bit file's order). */
static const uint8_t BYTE_LUT2[256] = {
    ['a'] = 0, ['A'] = 0, ['c'] = 1, ['C'] = 1, ['t'] = 2, ['T'] = 2, ['u'] = 2, ['U'] = 2,
    ['g'] = 3, ['G'] = 3, ['n'] = 4, ['N'] = 4,
};
static inline uint8_t byte_lut2(uint8_t c) { return BYTE_LUT2[c]; }

/* n_to_bits2.rs:25-33 */
static const uint8_t BITS_LUT2[5] = {'A', 'C', 'T', 'G', 'N'};

size_t cnt_oracle_words_for(size_t n_len) { return (n_len >> 5) + ((n_len & 31) ? 1 : 0); }
size_t cnt_oracle_words2_for(size_t n_len) { return n_len / 27 + ((n_len % 27) ? 1 : 0); }

/* ---- 2-bit codec ----------------------------------------------------------- */

/* n_to_bits.rs:34-47 */
int cnt_oracle_n_to_bits_lut(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words) {
    size_t words = cnt_oracle_words_for(n_len); /* :35 */
    if (out_words < words) return CNT_ORACLE_ECAP;
    if (words) memset(out, 0, words * sizeof(uint64_t)); /* vec![0u64; ..] :35 */
    for (size_t i = 0; i < n_len; i++) {                 /* :38 */
        size_t offset = i >> 5;                          /* :39 */
        unsigned shift = (unsigned)(i & 31) << 1;        /* :40 */
        out[offset] |= (uint64_t)byte_lut(n[i]) << shift; /* :41-42 */
    }
    return CNT_ORACLE_OK;
}

/* n_to_bits.rs:51-69 */
int cnt_oracle_bits_to_n_lut(const uint64_t *bits, size_t words, size_t len, uint8_t *out) {
    if (len > (words << 5)) return CNT_ORACLE_ELEN; /* :52-54 panic */
    for (size_t i = 0; i < len; i++) {              /* :60 */
        size_t offset = i >> 5;
        unsigned shift = (unsigned)(i & 31) << 1;
        uint64_t curr = bits[offset];
        out[i] = BITS_LUT[(curr >> shift) & 3];     /* :64 */
    }
    return CNT_ORACLE_OK;
}

/* What n_to_bits_{pext,shift,movemask,mul} compute on the 32-nt blocks: the two
 * bits under mask 0b110 of each byte (n_to_bits.rs:85,130,187-188,222), with
 * the <32-nt tail going through the LUT (n_to_bits.rs:109-111 and siblings). */
int cnt_oracle_n_to_bits_bitextract(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words) {
    size_t words = cnt_oracle_words_for(n_len);
    if (out_words < words) return CNT_ORACLE_ECAP;
    size_t end_idx = n_len >> 5;
    for (size_t w = 0; w < end_idx; w++) {
        uint64_t acc = 0;
        for (unsigned k = 0; k < 32; k++) acc |= (uint64_t)((n[(w << 5) + k] >> 1) & 3) << (2 * k);
        out[w] = acc;
    }
    if (n_len & 31) return cnt_oracle_n_to_bits_lut(n + (end_idx << 5), n_len & 31, out + end_idx, 1);
    return CNT_ORACLE_OK;
}

/* ---- 5-letter codec -------------------------------------------------------- */

/* n_to_bits2.rs:37-74 */
int cnt_oracle_n_to_bits2_lut(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words) {
    size_t words = cnt_oracle_words2_for(n_len); /* :38 */
    if (out_words < words) return CNT_ORACLE_ECAP;
    if (words) memset(out, 0, words * sizeof(uint64_t));
    size_t len = n_len / 3; /* :39 */
    for (size_t i = 0; i < len; i++) { /* :42 */
        size_t idx = i * 3;
        size_t res_offset = i / 9;
        unsigned res_shift = (unsigned)(i % 9) * 7;
        /* encoding = c*25 + b*5 + a  (:49-53); u8 arithmetic, max 4+20+100=124 */
        uint8_t a = byte_lut2(n[idx]);
        uint8_t b = (uint8_t)(byte_lut2(n[idx + 1]) * 5);
        uint8_t c = (uint8_t)(byte_lut2(n[idx + 2]) * 25);
        uint64_t encoding = (uint8_t)(a + b + c);
        out[res_offset] |= encoding << res_shift; /* :55 */
    }
    size_t leftover = n_len % 3; /* :58 */
    if (leftover > 0) {
        size_t idx = len * 3;
        size_t res_offset = len / 9;
        unsigned res_shift = (unsigned)(len % 9) * 7;
        uint8_t a = byte_lut2(n[idx]);
        uint8_t b = leftover >= 2 ? (uint8_t)(byte_lut2(n[idx + 1]) * 5) : 0; /* :66 */
        uint64_t encoding = (uint8_t)(a + b);
        out[res_offset] |= encoding << res_shift;
    }
    return CNT_ORACLE_OK;
}

/* n_to_bits2.rs:78-107.  The reference writes whole triplets into a buffer of
 * words*27 bytes and returns a Vec of length `len`; here only the first `len`
 * bytes are written (the observable result). */
int cnt_oracle_bits_to_n2_lut(const uint64_t *bits, size_t words, size_t len, uint8_t *out) {
    if (len > words * 27) return CNT_ORACLE_ELEN; /* :79-81 */
    size_t triplets = len / 3 + ((len % 3) ? 1 : 0); /* :83 */
    for (size_t i = 0; i < triplets; i++) {
        size_t idx = i * 3;
        size_t offset = i / 9;
        unsigned shift = (unsigned)(i % 9) * 7;
        uint64_t curr = (bits[offset] >> shift) & 0x7F; /* :95 */
        unsigned a = (unsigned)(curr % 5);
        unsigned b = (unsigned)((curr / 5) % 5);
        unsigned c = (unsigned)(curr / 25);
        /* c can reach 5 on words no encoder produces (curr 125..127); the
         * reference would read past its 5-entry LUT (UB).  Defined here as 'N'. */
        if (c > 4) c = 4;
        if (idx < len) out[idx] = BITS_LUT2[a];
        if (idx + 1 < len) out[idx + 1] = BITS_LUT2[b];
        if (idx + 2 < len) out[idx + 2] = BITS_LUT2[c];
    }
    return CNT_ORACLE_OK;
}

/* ---- generator + checksum (shared definition with the HIP library) -------- */

#define CNT_GOLDEN 0x9E3779B97F4A7C15ull

static uint64_t fmix64(uint64_t z) { /* splitmix64 finaliser (public domain, Vigna) */
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void cnt_oracle_fill_random_acgt(uint8_t *out, size_t first_nt, size_t n_len, uint64_t seed) {
    static const uint8_t ACGT[4] = {'A', 'C', 'G', 'T'};
    size_t w0 = first_nt >> 5;
    for (size_t i = 0; i < n_len;) {
        uint64_t w = w0 + (i >> 5);
        uint64_t r = fmix64(seed + (w + 1) * CNT_GOLDEN);
        size_t m = n_len - i < 32 ? n_len - i : 32;
        for (size_t k = 0; k < m; k++) out[i + k] = ACGT[(r >> (2 * k)) & 3];
        i += m;
    }
}

void cnt_oracle_fill_random_acgtn(uint8_t *out, size_t first_nt, size_t n_len, uint64_t seed) {
    static const uint8_t ACGT[4] = {'A', 'C', 'G', 'T'};
    size_t w0 = first_nt / 27;
    for (size_t i = 0; i < n_len;) {
        uint64_t w = w0 + i / 27;
        uint64_t r0 = fmix64(seed + (w + 1) * CNT_GOLDEN);
        uint64_t r1 = fmix64(r0 + CNT_GOLDEN);
        uint64_t r2 = fmix64(r1 + CNT_GOLDEN);
        size_t m = n_len - i < 27 ? n_len - i : 27;
        for (size_t k = 0; k < m; k++) {
            /* P(N) = 1/16: both 2-bit draws zero */
            int is_n = (((r1 >> (2 * k)) & 3) == 0) && (((r2 >> (2 * k)) & 3) == 0);
            out[i + k] = is_n ? 'N' : ACGT[(r0 >> (2 * k)) & 3];
        }
        i += m;
    }
}

uint64_t cnt_oracle_checksum_words(const uint64_t *w, size_t first_word, size_t words) {
    uint64_t s = 0;
    for (size_t i = 0; i < words; i++) s += fmix64(w[i] + (uint64_t)(first_word + i + 1) * CNT_GOLDEN);
    return s;
}

/* ---- packed-domain operations --------------------------------
 * NOT in the reference (README.md:20-25,45 only link to other projects for them), so
 * there are no reference vectors: parity for these four is pinned only by these
 * definitions, written independently of the HIP kernels (scalar, one nucleotide at a
 * time).  "parity unpinned" in the sense of the task statement. */
static unsigned code_at(const uint64_t *bits, size_t i) { return (unsigned)((bits[i >> 5] >> ((i & 31) << 1)) & 3); }
static void put_code(uint64_t *bits, size_t i, unsigned c) { bits[i >> 5] |= (uint64_t)c << ((i & 31) << 1); }

uint64_t cnt_oracle_hamming(const uint64_t *a, const uint64_t *b, size_t len) {
    uint64_t d = 0;
    for (size_t i = 0; i < len; i++) d += code_at(a, i) != code_at(b, i);
    return d;
}

/* A(0)<->T(2), C(1)<->G(3): code ^ 2 */
void cnt_oracle_complement(const uint64_t *bits, size_t len, uint64_t *out) {
    size_t words = cnt_oracle_words_for(len);
    if (words) memset(out, 0, words * 8);
    for (size_t i = 0; i < len; i++) put_code(out, i, code_at(bits, i) ^ 2u);
}

void cnt_oracle_reverse_complement(const uint64_t *bits, size_t len, uint64_t *out) {
    size_t words = cnt_oracle_words_for(len);
    if (words) memset(out, 0, words * 8);
    for (size_t i = 0; i < len; i++) put_code(out, i, code_at(bits, len - 1 - i) ^ 2u);
}

uint64_t cnt_oracle_validate(const uint8_t *n, size_t n_len, int allow_n) {
    uint64_t bad = 0;
    for (size_t i = 0; i < n_len; i++) {
        uint8_t c = n[i];
        int ok = c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'U' || c == 'a' || c == 'c' || c == 'g' || c == 't' ||
                 c == 'u' || (allow_n && (c == 'N' || c == 'n'));
        bad += !ok;
    }
    return bad;
}
