/* cnt_oracle_selftest.c -- stand-alone self-test of the oracle, meant to be run under
 * AddressSanitizer + UBSan (`make -C oracle asan`): the reference's known-answer vectors
 * (src/n_to_bits.rs:412-469, src/n_to_bits2.rs:274-298), ragged lengths right up to the ends
 * of exactly-sized heap buffers (so any over-read / over-write of the restatements or of the
 * SIMD ports trips ASan), and agreement between the scalar functions and the ports.
 * TEST INFRASTRUCTURE ONLY. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cnt_oracle.h"

static int fails = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            fails++;                                                       \
        }                                                                  \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

int main(void) {
    /* ---- KATs ------------------------------------------------------------------------ */
    const char *e1 = "ATCGATCGATCGATCGATCGATCGATCGATCG";
    uint64_t w[2];
    CHECK(cnt_oracle_n_to_bits_lut((const uint8_t *)e1, 32, w, 1) == 0 && w[0] == 0xD8D8D8D8D8D8D8D8ull);
    CHECK(cnt_oracle_n_to_bits_lut((const uint8_t *)"ATCG", 4, w, 1) == 0 && w[0] == 0xD8ull);
    uint8_t back[64];
    w[0] = 0xD8D8D8D8D8D8D8D8ull;
    CHECK(cnt_oracle_bits_to_n_lut(w, 1, 32, back) == 0 && memcmp(back, e1, 32) == 0);
    CHECK(cnt_oracle_bits_to_n_lut(w, 1, 33, back) == CNT_ORACLE_ELEN);
    const char *e3 = "ATCGNATCGNATCGNATCGNATCGNATCGNATCGN";
    CHECK(cnt_oracle_n_to_bits2_lut((const uint8_t *)e3, 35, w, 2) == 0 && w[0] == 0x36A45D1F46D48BA3ull && w[1] == 0x5D1F4ull);
    CHECK(cnt_oracle_n_to_bits2_lut((const uint8_t *)"ATCGN", 5, w, 1) == 0 && w[0] == 0xBA3ull);
    w[0] = 0x36A45D1F46D48BA3ull;
    w[1] = 0x5D1F4ull;
    CHECK(cnt_oracle_bits_to_n2_lut(w, 2, 35, back) == 0 && memcmp(back, e3, 35) == 0);

    /* ---- ragged lengths on exactly-sized heap buffers ------------------------------------ */
    const int simd = cnt_port_cpu_ok();
    static const char alpha[] = "ACGTUacgtu", alpha5[] = "ACGTNacgtnUu";
    for (size_t n_len = 0; n_len <= 200; n_len++) {
        uint8_t *n = (uint8_t *)malloc(n_len ? n_len : 1);
        uint8_t *n5 = (uint8_t *)malloc(n_len ? n_len : 1);
        for (size_t i = 0; i < n_len; i++) {
            n[i] = (uint8_t)alpha[rnd() % 10];
            n5[i] = (uint8_t)alpha5[rnd() % 12];
        }
        const size_t words = cnt_oracle_words_for(n_len), words5 = cnt_oracle_words2_for(n_len);
        uint64_t *a = (uint64_t *)malloc(words ? words * 8 : 8), *b = (uint64_t *)malloc(words ? words * 8 : 8);
        CHECK(cnt_oracle_n_to_bits_lut(n, n_len, a, words) == 0);
        CHECK(cnt_oracle_n_to_bits_bitextract(n, n_len, b, words) == 0 && memcmp(a, b, words * 8) == 0);
        if (simd) {
            CHECK(cnt_port_n_to_bits_pext(n, n_len, b, words) == 0 && memcmp(a, b, words * 8) == 0);
            CHECK(cnt_port_n_to_bits_shift(n, n_len, b, words) == 0 && memcmp(a, b, words * 8) == 0);
            CHECK(cnt_port_n_to_bits_movemask(n, n_len, b, words) == 0 && memcmp(a, b, words * 8) == 0);
            CHECK(cnt_port_n_to_bits_mul(n, n_len, b, words) == 0 && memcmp(a, b, words * 8) == 0);
        }
        uint8_t *d = (uint8_t *)malloc(n_len ? n_len : 1);
        CHECK(cnt_oracle_bits_to_n_lut(a, words, n_len, d) == 0);
        for (size_t i = 0; i < n_len; i++) {
            uint8_t up = (uint8_t)(n[i] & 0xDF);
            if (up == 'U') up = 'T';
            CHECK(d[i] == up);
        }
        if (simd) { /* the SIMD decoders store whole 32-byte blocks: give them exactly words*32 bytes, 32-aligned */
            uint8_t *blk = (uint8_t *)aligned_alloc(32, words ? words * 32 : 32);
            CHECK(cnt_port_bits_to_n_shuffle(a, words, n_len, blk) == 0 && memcmp(blk, d, n_len) == 0);
            CHECK(cnt_port_bits_to_n_pdep(a, words, n_len, blk) == 0 && memcmp(blk, d, n_len) == 0);
            CHECK(cnt_port_bits_to_n_clmul(a, words, n_len, blk) == 0 && memcmp(blk, d, n_len) == 0);
            free(blk);
        }
        uint64_t *a5 = (uint64_t *)malloc(words5 ? words5 * 8 : 8), *b5 = (uint64_t *)malloc(words5 ? words5 * 8 : 8);
        CHECK(cnt_oracle_n_to_bits2_lut(n5, n_len, a5, words5) == 0);
        if (simd) CHECK(cnt_port_n_to_bits2_pext(n5, n_len, b5, words5) == 0 && memcmp(a5, b5, words5 * 8) == 0);
        uint8_t *d5 = (uint8_t *)malloc(n_len ? n_len : 1);
        CHECK(cnt_oracle_bits_to_n2_lut(a5, words5, n_len, d5) == 0);
        if (simd) {
            uint8_t *blk = (uint8_t *)malloc(words5 * 27 + 5 + 1);
            CHECK(cnt_port_bits_to_n2_pdep(a5, words5, n_len, blk) == 0 && memcmp(blk, d5, n_len) == 0);
            free(blk);
        }
        /* packed-domain definitions: involutions */
        uint64_t *c1 = (uint64_t *)malloc(words ? words * 8 : 8), *c2 = (uint64_t *)malloc(words ? words * 8 : 8);
        cnt_oracle_complement(a, n_len, c1);
        cnt_oracle_complement(c1, n_len, c2);
        CHECK(memcmp(c2, a, words * 8) == 0);
        cnt_oracle_reverse_complement(a, n_len, c1);
        cnt_oracle_reverse_complement(c1, n_len, c2);
        CHECK(memcmp(c2, a, words * 8) == 0);
        CHECK(cnt_oracle_hamming(a, a, n_len) == 0);
        cnt_oracle_complement(a, n_len, c1);
        CHECK(cnt_oracle_hamming(a, c1, n_len) == n_len);
        CHECK(cnt_oracle_validate(n, n_len, 0) == 0);
        free(n); free(n5); free(a); free(b); free(d); free(a5); free(b5); free(d5); free(c1); free(c2);
    }
    /* generator / checksum touch exactly their ranges */
    uint8_t *g = (uint8_t *)malloc(1000);
    cnt_oracle_fill_random_acgt(g, 64, 1000, 1);
    cnt_oracle_fill_random_acgtn(g, 27, 1000, 1);
    free(g);
    if (fails) printf("oracle selftest: %d FAILURES\n", fails);
    else printf("oracle selftest ok (simd ports %s)\n", simd ? "checked" : "skipped: CPU lacks AVX2/BMI2/PCLMUL");
    return fails ? 1 : 0;
}
