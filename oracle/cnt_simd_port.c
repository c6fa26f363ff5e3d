/*
 * cnt_simd_port.c -- C/<immintrin.h> ports of the reference's x86 SIMD codecs.
 * TEST INFRASTRUCTURE ONLY (see cnt_oracle.h): this is the "reference's own
 * AVX2/BMI2 path" that bench.py times on the GPU box's host cores
 * (cpu_baseline.kind = "port" -- the reference is Rust and cannot be built in
 * this image), and a second opinion on the scalar oracle in tests/.
 *
 * Rust's std::arch::x86_64 intrinsics are the Intel intrinsics one-to-one, so
 * each function below issues the same instruction sequence as the cited
 * reference lines.  Differences are confined to memory ownership: outputs are
 * caller-allocated (the reference allocates a Vec inside each call; bench.py's
 * reference-faithful row allocates inside the timed region to match
 * benches/bench_n_to_bits.rs:6-7).
 *
 * Compiled for a generic x86-64 with per-function target attributes, so the .so
 * built in the build container still loads on the GPU box; cnt_port_cpu_ok()
 * reports whether the running CPU has AVX2+BMI2+PCLMULQDQ.
 */
#include "cnt_oracle.h"

#include <immintrin.h>
#include <string.h>

#define CNT_SIMD __attribute__((target("avx2,bmi2,pclmul,ssse3")))

int cnt_port_cpu_ok(void) {
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("pclmul");
}

/* Tail rule shared by all SIMD encoders (n_to_bits.rs:109-111,160-162,201-203,
 * 253-255): the last partial word comes from the scalar LUT. */
static int encode_tail(const uint8_t *n, size_t n_len, uint64_t *out, size_t end_idx) {
    if (n_len & 31) return cnt_oracle_n_to_bits_lut(n + (end_idx << 5), n_len & 31, out + end_idx, 1);
    return CNT_ORACLE_OK;
}

/* n_to_bits.rs:80-115 */
CNT_SIMD int cnt_port_n_to_bits_pext(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words) {
    if (!cnt_port_cpu_ok()) return CNT_ORACLE_ECPU;
    size_t end_idx = n_len >> 5;
    if (out_words < cnt_oracle_words_for(n_len)) return CNT_ORACLE_ECAP;
    const uint64_t ascii_mask = 0x0606060606060606ull; /* :85 */
    for (size_t i = 0; i < end_idx; i++) {
        uint64_t q[4];
        memcpy(q, n + (i << 5), 32); /* loadu + union read, :96 */
        uint64_t a = _pext_u64(q[0], ascii_mask);
        uint64_t b = _pext_u64(q[1], ascii_mask);
        uint64_t c = _pext_u64(q[2], ascii_mask);
        uint64_t d = _pext_u64(q[3], ascii_mask);
        out[i] = a | (b << 16) | (c << 32) | (d << 48); /* :106 */
    }
    return encode_tail(n, n_len, out, end_idx);
}

/* n_to_bits.rs:121-166 */
CNT_SIMD int cnt_port_n_to_bits_shift(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words) {
    if (!cnt_port_cpu_ok()) return CNT_ORACLE_ECPU;
    size_t end_idx = n_len >> 5;
    if (out_words < cnt_oracle_words_for(n_len)) return CNT_ORACLE_ECAP;
    const __m256i ascii_mask = _mm256_set1_epi8(0x06);
    const __m256i shuffle_mask = _mm256_set_epi32(-1, -1, -1, 0x0C080400, -1, -1, -1, 0x0C080400);
    for (size_t i = 0; i < end_idx; i++) {
        __m256i v = _mm256_loadu_si256((const __m256i *)(n + (i << 5)));
        v = _mm256_and_si256(v, ascii_mask);
        __m256i a = _mm256_srli_epi16(v, 1);
        __m256i b = _mm256_srli_epi16(v, 8 - 2 + 1);
        a = _mm256_or_si256(a, b);
        b = _mm256_srli_epi32(a, 16 - 4);
        v = _mm256_or_si256(a, b);
        v = _mm256_shuffle_epi8(v, shuffle_mask);
        out[i] = (uint64_t)_mm256_extract_epi64(v, 0) | ((uint64_t)_mm256_extract_epi64(v, 2) << 32); /* :157 */
    }
    return encode_tail(n, n_len, out, end_idx);
}

/* n_to_bits.rs:172-207 -- the reference's fastest encoder */
CNT_SIMD int cnt_port_n_to_bits_movemask(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words) {
    if (!cnt_port_cpu_ok()) return CNT_ORACLE_ECPU;
    size_t end_idx = n_len >> 5;
    if (out_words < cnt_oracle_words_for(n_len)) return CNT_ORACLE_ECAP;
    for (size_t i = 0; i < end_idx; i++) {
        __m256i v = _mm256_loadu_si256((const __m256i *)(n + (i << 5)));
        v = _mm256_permute4x64_epi64(v, 0xD8); /* 0b11011000, :184 */
        __m256i lo = _mm256_slli_epi64(v, 6);
        __m256i hi = _mm256_slli_epi64(v, 5);
        __m256i a = _mm256_unpackhi_epi8(lo, hi);
        __m256i b = _mm256_unpacklo_epi8(lo, hi);
        uint64_t am = (uint32_t)_mm256_movemask_epi8(a);
        uint64_t bm = (uint32_t)_mm256_movemask_epi8(b);
        out[i] = (am << 32) | bm; /* :198 */
    }
    return encode_tail(n, n_len, out, end_idx);
}

/* n_to_bits.rs:213-259 */
CNT_SIMD int cnt_port_n_to_bits_mul(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words) {
    if (!cnt_port_cpu_ok()) return CNT_ORACLE_ECPU;
    size_t end_idx = n_len >> 5;
    if (out_words < cnt_oracle_words_for(n_len)) return CNT_ORACLE_ECAP;
    const __m256i ascii_mask = _mm256_set1_epi8(0x06);
    uint32_t m = 0; /* :223-231 */
    m |= 1u << (32 - 8 + 0 - 1);
    m |= 1u << (32 - 16 + 2 - 1);
    m |= 1u << (32 - 24 + 4 - 1);
    m |= 1u << (32 - 32 + 6 - 1);
    const __m256i mul_mask = _mm256_set1_epi32((int)m);
    const __m256i shuffle_mask = _mm256_set_epi32(-1, -1, -1, 0x0F0B0703, -1, -1, -1, 0x0F0B0703);
    for (size_t i = 0; i < end_idx; i++) {
        __m256i v = _mm256_loadu_si256((const __m256i *)(n + (i << 5)));
        v = _mm256_and_si256(v, ascii_mask);
        v = _mm256_mullo_epi32(v, mul_mask);
        v = _mm256_shuffle_epi8(v, shuffle_mask);
        out[i] = (uint64_t)_mm256_extract_epi64(v, 0) | ((uint64_t)_mm256_extract_epi64(v, 2) << 32);
    }
    return encode_tail(n, n_len, out, end_idx);
}

#define LUT_I32 ((int)('A' | ('C' << 8) | ('T' << 16) | ('G' << 24)))

/* n_to_bits.rs:265-303 -- the reference's fastest decoder.  Stores words*32
 * bytes with aligned 32-byte stores, like the reference. */
CNT_SIMD int cnt_port_bits_to_n_shuffle(const uint64_t *bits, size_t words, size_t len, uint8_t *out) {
    if (!cnt_port_cpu_ok()) return CNT_ORACLE_ECPU;
    if (len > (words << 5)) return CNT_ORACLE_ELEN;
    const __m256i shuffle_mask = _mm256_set_epi32(0x07070707, 0x06060606, 0x05050505, 0x04040404, 0x03030303, 0x02020202,
                                                  0x01010101, 0x00000000);
    const __m256i lo_mask = _mm256_set1_epi16(0x0C03); /* 0b0000110000000011 */
    const __m256i lut = _mm256_set_epi32('G', 'T', 'C', LUT_I32, 'G', 'T', 'C', LUT_I32);
    __m256i *ptr = (__m256i *)out;
    for (size_t i = 0; i < words; i++) {
        __m256i v = _mm256_set1_epi64x((long long)bits[i]);
        __m256i v1 = _mm256_shuffle_epi8(v, shuffle_mask);
        __m256i v2 = _mm256_srli_epi16(v1, 4);
        v = _mm256_blend_epi16(v1, v2, 0xAA);
        v = _mm256_and_si256(v, lo_mask);
        v = _mm256_shuffle_epi8(lut, v);
        _mm256_store_si256(ptr + i, v);
    }
    return CNT_ORACLE_OK;
}

/* n_to_bits.rs:309-340 */
CNT_SIMD int cnt_port_bits_to_n_pdep(const uint64_t *bits, size_t words, size_t len, uint8_t *out) {
    if (!cnt_port_cpu_ok()) return CNT_ORACLE_ECPU;
    if (len > (words << 5)) return CNT_ORACLE_ELEN;
    const uint64_t scatter_mask = 0x0303030303030303ull;
    const __m256i lut = _mm256_set_epi32(0, 0, 0, LUT_I32, 0, 0, 0, LUT_I32);
    __m256i *ptr = (__m256i *)out;
    for (size_t i = 0; i < words; i++) {
        uint64_t curr = bits[i];
        long long a = (long long)_pdep_u64(curr, scatter_mask);
        long long b = (long long)_pdep_u64(curr >> 16, scatter_mask);
        long long c = (long long)_pdep_u64(curr >> 32, scatter_mask);
        long long d = (long long)_pdep_u64(curr >> 48, scatter_mask);
        __m256i v = _mm256_set_epi64x(d, c, b, a);
        v = _mm256_shuffle_epi8(lut, v);
        _mm256_store_si256(ptr + i, v);
    }
    return CNT_ORACLE_OK;
}

/* n_to_bits.rs:346-404 */
CNT_SIMD int cnt_port_bits_to_n_clmul(const uint64_t *bits, size_t words, size_t len, uint8_t *out) {
    if (!cnt_port_cpu_ok()) return CNT_ORACLE_ECPU;
    if (len > (words << 5)) return CNT_ORACLE_ELEN;
    const __m128i lo_shuffle_mask = _mm_set_epi32((int)0xFFFFFF03u, (int)0xFFFFFF02u, (int)0xFFFFFF01u, (int)0xFFFFFF00u);
    const __m128i hi_shuffle_mask = _mm_set_epi32((int)0xFFFFFF07u, (int)0xFFFFFF06u, (int)0xFFFFFF05u, (int)0xFFFFFF04u);
    uint64_t m = 0; /* :357-365 */
    m |= 1ull << (0 - 0);
    m |= 1ull << (8 - 2);
    m |= 1ull << (16 - 4);
    m |= 1ull << (24 - 6);
    const __m128i mul_mask = _mm_set_epi64x(0, (long long)m);
    const __m128i lo_mask = _mm_set1_epi8(0x03);
    const __m128i lut = _mm_set1_epi32(LUT_I32);
    __m128i *ptr = (__m128i *)out;
    for (size_t i = 0; i < words; i++) {
        __m128i v = _mm_set1_epi64x((long long)bits[i]);
        __m128i lo_v = _mm_shuffle_epi8(v, lo_shuffle_mask);
        __m128i hi_v = _mm_shuffle_epi8(v, hi_shuffle_mask);
        __m128i lo_v1 = _mm_clmulepi64_si128(lo_v, mul_mask, 0x00);
        __m128i lo_v2 = _mm_clmulepi64_si128(lo_v, mul_mask, 0x0F);
        __m128i hi_v1 = _mm_clmulepi64_si128(hi_v, mul_mask, 0x00);
        __m128i hi_v2 = _mm_clmulepi64_si128(hi_v, mul_mask, 0x0F);
        lo_v = _mm_castps_si128(_mm_movelh_ps(_mm_castsi128_ps(lo_v1), _mm_castsi128_ps(lo_v2)));
        hi_v = _mm_castps_si128(_mm_movelh_ps(_mm_castsi128_ps(hi_v1), _mm_castsi128_ps(hi_v2)));
        lo_v = _mm_and_si128(lo_v, lo_mask);
        hi_v = _mm_and_si128(hi_v, lo_mask);
        lo_v = _mm_shuffle_epi8(lut, lo_v);
        hi_v = _mm_shuffle_epi8(lut, hi_v);
        _mm_store_si128(ptr + (i << 1), lo_v);
        _mm_store_si128(ptr + (i << 1) + 1, hi_v);
    }
    return CNT_ORACLE_OK;
}

/* ---- 5-letter codec -------------------------------------------------------- */

/* n_to_bits2.rs:118-189 */
CNT_SIMD int cnt_port_n_to_bits2_pext(const uint8_t *n, size_t n_len, uint64_t *out, size_t out_words) {
    if (!cnt_port_cpu_ok()) return CNT_ORACLE_ECPU;
    size_t end_idx = n_len < 5 ? 0 : (n_len - 5) / 27; /* 32-B loads at stride 27 over-read 5 B, :120 */
    size_t len = cnt_oracle_words2_for(n_len);
    if (out_words < len) return CNT_ORACLE_ECAP;
    long long lutq = 0; /* :127-136: table indexed by the low 3 bits of the ASCII code */
    lutq |= 0ll << ((('A') & 7) << 3);
    lutq |= 1ll << ((('C') & 7) << 3);
    lutq |= 2ll << ((('T') & 7) << 3);
    lutq |= 2ll << ((('U') & 7) << 3);
    lutq |= 3ll << ((('G') & 7) << 3);
    lutq |= 4ll << ((('N') & 7) << 3);
    const __m256i lut = _mm256_set1_epi64x(lutq);
    const __m256i permute_mask = _mm256_set_epi32(6, 5, 4, 3, 3, 2, 1, 0);
    const __m256i lo_shuffle_mask =
        _mm256_set_epi16(-1, -1, -1, -1, (short)0xFF1C, (short)0xFF19, (short)0xFF16, (short)0xFF13, -1, -1, -1,
                         (short)0xFF0C, (short)0xFF09, (short)0xFF06, (short)0xFF03, (short)0xFF00);
    const __m256i hi_shuffle_mask =
        _mm256_set_epi16(-1, -1, -1, -1, (short)0x1E1D, (short)0x1B1A, (short)0x1817, (short)0x1514, -1, -1, -1,
                         (short)0x0E0D, (short)0x0B0A, (short)0x0807, (short)0x0504, (short)0x0201);
    const __m256i mul_25_5 = _mm256_set1_epi16(0x1905);
    const uint64_t pack_right_mask = 0x007F007F007F007Full;
    const uint8_t *ptr = n;
    for (size_t i = 0; i < end_idx; i++) {
        __m256i v = _mm256_loadu_si256((const __m256i *)ptr);
        v = _mm256_shuffle_epi8(lut, v);
        v = _mm256_permutevar8x32_epi32(v, permute_mask);
        __m256i a = _mm256_shuffle_epi8(v, lo_shuffle_mask);
        __m256i b = _mm256_shuffle_epi8(v, hi_shuffle_mask);
        b = _mm256_maddubs_epi16(b, mul_25_5);
        __m256i s = _mm256_add_epi16(a, b);
        uint64_t q0 = (uint64_t)_mm256_extract_epi64(s, 0);
        uint64_t q1 = (uint64_t)_mm256_extract_epi64(s, 1);
        uint64_t q2 = (uint64_t)_mm256_extract_epi64(s, 2);
        uint64_t pa = _pext_u64(q0, pack_right_mask);
        uint64_t pc = _pext_u64(q2, pack_right_mask);
        out[i] = pa | (q1 << 28) | (pc << 35); /* :174 */
        ptr += 27;
    }
    if (end_idx < len) /* :179-185 */
        return cnt_oracle_n_to_bits2_lut(n + end_idx * 27, n_len - end_idx * 27, out + end_idx, len - end_idx);
    return CNT_ORACLE_OK;
}

/* n_to_bits2.rs:196-268.  `out` must hold words*27 + 5 bytes (:202). */
CNT_SIMD int cnt_port_bits_to_n2_pdep(const uint64_t *bits, size_t words, size_t len, uint8_t *out) {
    if (!cnt_port_cpu_ok()) return CNT_ORACLE_ECPU;
    if (len > words * 27) return CNT_ORACLE_ELEN;
    const uint64_t deposit_mask = 0x7F7F7F7F7F7F7F7Full;
    const __m256i shuffle_mask =
        _mm256_set_epi16(-1, -1, -1, (short)0xFF04, (short)0xFF03, (short)0xFF02, (short)0xFF01, (short)0xFF00, -1, -1, -1,
                         -1, (short)0xFF03, (short)0xFF02, (short)0xFF01, (short)0xFF00);
    const __m256i mul5 = _mm256_set1_epi16(5);
    const __m256i div5 = _mm256_set1_epi16((short)((1u << 16) / 5 + 1));
    const __m256i div25 = _mm256_set1_epi16((short)((1u << 16) / 25 + 1));
    const __m256i a_shuffle_mask = _mm256_set_epi64x((long long)0xFFFFFF08FFFF06FFull, (long long)0xFF04FFFF02FFFF00ull,
                                                     (long long)0xFFFFFF08FFFF06FFull, (long long)0xFF04FFFF02FFFF00ull);
    const __m256i b_shuffle_mask = _mm256_set_epi64x((long long)0xFFFF08FFFF06FFFFull, (long long)0x04FFFF02FFFF00FFull,
                                                     (long long)0xFFFF08FFFF06FFFFull, (long long)0x04FFFF02FFFF00FFull);
    const __m256i c_shuffle_mask = _mm256_set_epi64x((long long)0xFF08FFFF06FFFF04ull, (long long)0xFFFF02FFFF00FFFFull,
                                                     (long long)0xFF08FFFF06FFFF04ull, (long long)0xFFFF02FFFF00FFFFull);
    const __m256i permute_mask = _mm256_set_epi32(7, 7, 6, 5, 4, 2, 1, 0);
    long long lutq = 0;
    lutq |= (long long)'A' << 0;
    lutq |= (long long)'C' << 8;
    lutq |= (long long)'T' << 16;
    lutq |= (long long)'G' << 24;
    lutq |= (long long)'N' << 32;
    const __m256i lut = _mm256_set1_epi64x(lutq);
    uint8_t *ptr = out;
    for (size_t i = 0; i < words; i++) {
        long long curr = (long long)bits[i];
        long long a = (long long)_pdep_u64((uint64_t)curr, deposit_mask);
        long long b = ((curr >> 56) << 32) | (a >> 32);
        __m256i v = _mm256_set_epi64x(0, b, 0, a);
        v = _mm256_shuffle_epi8(v, shuffle_mask);
        __m256i v_rem5 = _mm256_mullo_epi16(v, div5);
        __m256i v_rem25 = _mm256_mullo_epi16(v, div25);
        __m256i va = _mm256_mulhi_epu16(v_rem5, mul5);
        __m256i vb = _mm256_mulhi_epu16(v_rem25, mul5);
        __m256i vc = _mm256_mulhi_epu16(v, div25);
        va = _mm256_shuffle_epi8(va, a_shuffle_mask);
        vb = _mm256_shuffle_epi8(vb, b_shuffle_mask);
        vc = _mm256_shuffle_epi8(vc, c_shuffle_mask);
        __m256i abc = _mm256_or_si256(_mm256_or_si256(va, vb), vc);
        v = _mm256_permutevar8x32_epi32(abc, permute_mask);
        v = _mm256_shuffle_epi8(lut, v);
        _mm256_storeu_si256((__m256i *)ptr, v);
        ptr += 27;
    }
    return CNT_ORACLE_OK;
}

/* ---- reference-faithful timing row -------------------------------------------
 * benches/bench_n_to_bits.rs:6-7: "all functions must allocate memory for its
 * output data" -- every timed call of the reference allocates (and drops) its
 * result.  This helper times `iters` calls of one function with malloc/free of
 * the output inside the timed region and returns seconds per call, so bench.py
 * can print rows that sit beside README.md:344-366 (40 000 nt, one thread).
 * fn: 0 lut 1 pext 2 shift 3 movemask 4 mul 5 memcpy | 10 lut 11 shuffle 12 pdep 13 clmul
 *     20 n_to_bits2_lut 21 n_to_bits2_pext | 30 bits_to_n2_lut 31 bits_to_n2_pdep   (5-letter codec,
 *     benches/bench_n_to_bits.rs:31-32,59-60; n_len is always the nucleotide count) */
#include <stdlib.h>
#include <time.h>

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double cnt_port_time_alloc_inclusive(int fn, const void *in, size_t n_len, int iters) {
    typedef int (*enc_t)(const uint8_t *, size_t, uint64_t *, size_t);
    typedef int (*dec_t)(const uint64_t *, size_t, size_t, uint8_t *);
    static const enc_t encs[5] = {cnt_oracle_n_to_bits_lut, cnt_port_n_to_bits_pext, cnt_port_n_to_bits_shift,
                                  cnt_port_n_to_bits_movemask, cnt_port_n_to_bits_mul};
    static const dec_t decs[4] = {cnt_oracle_bits_to_n_lut, cnt_port_bits_to_n_shuffle, cnt_port_bits_to_n_pdep,
                                  cnt_port_bits_to_n_clmul};
    const size_t words = cnt_oracle_words_for(n_len);
    volatile uint64_t sink = 0;
    double t0 = now_s();
    for (int i = 0; i < iters; i++) {
        if (fn >= 0 && fn < 5) {
            uint64_t *out = (uint64_t *)malloc(words * 8 + 8);
            encs[fn]((const uint8_t *)in, n_len, out, words);
            sink += out[words - 1];
            free(out);
        } else if (fn == 5) {
            uint8_t *out = (uint8_t *)malloc(n_len);
            memcpy(out, in, n_len);
            sink += out[n_len - 1];
            free(out);
        } else if (fn >= 10 && fn < 14) {
            uint8_t *out = (uint8_t *)aligned_alloc(32, words * 32);
            decs[fn - 10]((const uint64_t *)in, words, n_len, out);
            sink += out[n_len - 1];
            free(out);
        } else if (fn == 20 || fn == 21) {
            const size_t w2 = cnt_oracle_words2_for(n_len);
            uint64_t *out = (uint64_t *)malloc(w2 * 8 + 8);
            (fn == 20 ? cnt_oracle_n_to_bits2_lut : cnt_port_n_to_bits2_pext)((const uint8_t *)in, n_len, out, w2);
            sink += out[w2 - 1];
            free(out);
        } else if (fn == 30 || fn == 31) {
            const size_t w2 = cnt_oracle_words2_for(n_len);
            uint8_t *out = (uint8_t *)malloc(w2 * 27 + 32); /* n_to_bits2.rs:202: over-allocated for the 32-B stores */
            (fn == 30 ? cnt_oracle_bits_to_n2_lut : cnt_port_bits_to_n2_pdep)((const uint64_t *)in, w2, n_len, out);
            sink += out[n_len - 1];
            free(out);
        } else {
            return -1.0;
        }
    }
    (void)sink;
    return (now_s() - t0) / (double)iters;
}
