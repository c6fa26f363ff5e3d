/* c_link_check.c -- include/cute_nt.h is a C header and libcute_nt_hip.so has C linkage: a plain C11 program
 * (gcc, no C++ anywhere) links against it and walks the entry points that need no device.  Built and run by
 * tests/test_host_mirrors.py on the CPU box (no GPU there: compute calls must answer CNT_ENODEV, not crash). */
#include <stdio.h>
#include <string.h>

#include "../include/cute_nt.h"

int main(void) {
    size_t lo = 0, hi = 0;
    int count = -1;
    uint8_t n[64];
    uint64_t out[2] = {0, 0};
    memset(n, 'A', sizeof n);
    if (cnt_abi_version() != CNT_ABI_VERSION) return 1;
    if (cnt_words_for(33) != 2 || cnt_words2_for(28) != 2) return 2;
    if (strcmp(cnt_strerror(CNT_ELEN), "The length is greater than the number of nucleotides!") != 0) return 3;
    if (cnt_shard_range((size_t)1 << 35, 8, 3, 32, &lo, &hi) != CNT_OK || lo != (size_t)3 << 32 || hi != (size_t)4 << 32) return 4;
    if (cnt_n_to_bits(n, 64, out, 1) != CNT_ECAP) return 5;              /* argument errors need no device */
    if (cnt_bits_to_n(out, 1, 33, n) != CNT_ELEN) return 6;
    if (cnt_n_to_bits(NULL, 0, NULL, 0) != CNT_OK) return 7;               /* empty in -> empty out */
    if (cnt_n_to_bits(n, 32, (uint64_t *)(n + 24), 1) != CNT_EINVAL) return 14;  /* input and output never share memory */
    if (cnt_device_count(&count) != CNT_OK) return 8;
    if (count == 0 && cnt_n_to_bits(n, 64, out, 2) != CNT_ENODEV) return 9;  /* no CPU fallback */
    {   /* the enqueue-only multi-GPU tier from C: no device -> no queue; unknown handles are refused, not dereferenced */
        void *q = (void *)out;
        int shards = -1;
        if (count == 0 && (cnt_sharded_dev_open(1, CNT_QUEUE_TIMED, &q) != CNT_ENODEV || q != NULL)) return 10;
        if (cnt_sharded_dev_wait((void *)n, NULL) != CNT_EINVAL || cnt_sharded_dev_close((void *)n) != CNT_EINVAL) return 11;
        if (cnt_sharded_dev_shards((void *)n, &shards) != CNT_EINVAL || cnt_sharded_dev_op_ms((void *)n, 0, NULL) != CNT_EINVAL) return 12;
        if (cnt_set_tuning("encode", 0) != CNT_EINVAL) return 13;            /* the product library selects nothing at run time */
    }
    printf("c link ok: abi %d, %d device(s)\n", cnt_abi_version(), count);
    return 0;
}
