// codec5_launch.hpp -- kernel-variant tables + launchers of the 5-letter codec (see
// codec2_launch.hpp for the conventions: variant 0 = shipped default, the rest selectable
// through cnt_set_tuning("encode2" / "decode2", i) for A/B runs).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codec2_launch.hpp"
#include "codec5_kernels.hpp"

namespace cnt {

// tile_nt = nucleotides per WAVE tile (WPL * 1728); a workgroup takes `waves` of them
constexpr VariantDesc kEncode2Variants[] = {
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=sc1, 15 wg/CU", 2 * kWaveBytes5, 64, 15},  // 0: default
#ifdef CNT_LAB_VARIANTS
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=nt st=sc1", 4 * kWaveBytes5, 64, 0},   // 1
    {"wave-tiled 2 words/lane, 2 waves/wg, ld=nt st=sc1", 2 * kWaveBytes5, 64, 0},  // 2
    {"wave-tiled 2 words/lane, 4 waves/wg, ld=nt st=sc1", 2 * kWaveBytes5, 64, 0},  // 3
    {"wave-tiled 1 word/lane, 1 wave/wg, ld=nt st=sc1 (tiles split cache lines)", kWaveBytes5, 64, 0},  // 4
    {"wave-tiled 2 words/lane, 1 wave/wg, plain", 2 * kWaveBytes5, 64, 0},          // 5
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=sc1", 2 * kWaveBytes5, 64, 0},   // 6: as 0 without the residency cap
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=sc1, 20 wg/CU", 2 * kWaveBytes5, 64, 20},  // 7
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=sc1, 16 wg/CU", 2 * kWaveBytes5, 64, 16},  // 8
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-pairs, ld=nt st=sc1, 24 wg/CU", 2 * kWaveBytes5, 64, 24},  // 9
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-pairs, ld=nt st=sc0|sc1|nt, 24 wg/CU", 2 * kWaveBytes5, 64, 24},  // 10
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=sc1, 12 wg/CU", 2 * kWaveBytes5, 64, 12},  // 11
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=sc1, 14 wg/CU", 2 * kWaveBytes5, 64, 14},  // 12
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=sc1, 18 wg/CU", 2 * kWaveBytes5, 64, 18},  // 13
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=sc1, 10 wg/CU", 2 * kWaveBytes5, 64, 10},  // 14
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=sc1, 24 wg/CU", 2 * kWaveBytes5, 64, 24},  // 15: the default before the caps were re-swept
    // round 2: 4 words per lane (6912 B of ASCII per wave) was only ever measured uncapped (variant 1); the fused 2-bit
    // kernel's sweep says fat one-wave tiles want a LOW residency cap
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=nt st=sc1, 7 wg/CU", 4 * kWaveBytes5, 64, 7},    // 16
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=nt st=sc1, 8 wg/CU", 4 * kWaveBytes5, 64, 8},    // 17
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=nt st=sc1, 9 wg/CU", 4 * kWaveBytes5, 64, 9},    // 18
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=nt st=sc1, 10 wg/CU", 4 * kWaveBytes5, 64, 10},  // 19
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=nt st=sc1, 12 wg/CU", 4 * kWaveBytes5, 64, 12},  // 20
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=nt st=sc0|sc1|nt, 8 wg/CU", 4 * kWaveBytes5, 64, 8},  // 21
    // round 3: the default shape with the store / load policies that were only ever measured together with other changes
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=sc0|sc1|nt, 15 wg/CU", 2 * kWaveBytes5, 64, 15},  // 22
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=sc1|nt, 15 wg/CU", 2 * kWaveBytes5, 64, 15},      // 23
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=nt, 15 wg/CU", 2 * kWaveBytes5, 64, 15},          // 24
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=sc1|nt st=sc1, 15 wg/CU", 2 * kWaveBytes5, 64, 15},     // 25
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=plain st=sc1, 15 wg/CU", 2 * kWaveBytes5, 64, 15},      // 26
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=nt st=sc0|sc1|nt, 14 wg/CU", 2 * kWaveBytes5, 64, 14},  // 27
    // round 4: pipelined -- one wave takes K consecutive tiles and loads tile i+1 before it touches tile i (codec5_kernels.hpp)
    {"pipelined K=2, ld=nt st=sc1, 15 wg/CU", 2 * kWaveBytes5, 64, 15},  // 28
    {"pipelined K=2, ld=nt st=sc1, 12 wg/CU", 2 * kWaveBytes5, 64, 12},  // 29
    {"pipelined K=2, ld=nt st=sc1, 10 wg/CU", 2 * kWaveBytes5, 64, 10},  // 30
    {"pipelined K=4, ld=nt st=sc1, 15 wg/CU", 2 * kWaveBytes5, 64, 15},  // 31
    {"pipelined K=4, ld=nt st=sc1, 12 wg/CU", 2 * kWaveBytes5, 64, 12},  // 32
    {"pipelined K=4, ld=nt st=sc1, 10 wg/CU", 2 * kWaveBytes5, 64, 10},  // 33
    {"pipelined K=8, ld=nt st=sc1, 12 wg/CU", 2 * kWaveBytes5, 64, 12},  // 34
    {"pipelined K=8, ld=nt st=sc1, 8 wg/CU", 2 * kWaveBytes5, 64, 8},    // 35
    {"pipelined K=2, ld=nt st=sc1, 20 wg/CU", 2 * kWaveBytes5, 64, 20},  // 36
    {"pipelined K=4, ld=nt st=sc0|sc1|nt, 12 wg/CU", 2 * kWaveBytes5, 64, 12},  // 37
    // multi-wave workgroups were only ever measured uncapped (variants 2, 3): four waves = 4 KiB of packed output per workgroup
    {"wave-tiled 2 words/lane, 4 waves/wg, ld=nt st=sc1, 3 wg/CU", 2 * kWaveBytes5, 64, 3},  // 38
    {"wave-tiled 2 words/lane, 4 waves/wg, ld=nt st=sc1, 4 wg/CU", 2 * kWaveBytes5, 64, 4},  // 39
    {"wave-tiled 2 words/lane, 4 waves/wg, ld=nt st=sc1, 5 wg/CU", 2 * kWaveBytes5, 64, 5},  // 40
    {"wave-tiled 2 words/lane, 2 waves/wg, ld=nt st=sc1, 7 wg/CU", 2 * kWaveBytes5, 64, 7},  // 41
    {"wave-tiled 2 words/lane, 2 waves/wg, ld=nt st=sc1, 8 wg/CU", 2 * kWaveBytes5, 64, 8},  // 42
    // 32 tiles = 27 whole 4-KiB pages of ASCII: the first group size at which every XCD turn reads whole pages
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-32s (27 pages per turn), ld=nt st=sc1, 15 wg/CU", 2 * kWaveBytes5, 64, 15},  // 43
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-32s, ld=nt st=sc1, 12 wg/CU", 2 * kWaveBytes5, 64, 12},  // 44
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-32s, ld=nt st=sc1, 18 wg/CU", 2 * kWaveBytes5, 64, 18},  // 45
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-quads, ld=nt st=sc1, 15 wg/CU", 2 * kWaveBytes5, 64, 15},  // 46: 4 KiB of packed OUTPUT per turn
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-8s, ld=nt st=sc1, 15 wg/CU", 2 * kWaveBytes5, 64, 15},    // 47
    // branch-free twin of variant 0 (the pipelined kernel with K = 1: all four loads and LDS writes without a lane mask)
    {"branch-free K=1, ld=nt st=sc1, 15 wg/CU", 2 * kWaveBytes5, 64, 15},  // 48
    {"branch-free K=1, ld=nt st=sc1, 16 wg/CU", 2 * kWaveBytes5, 64, 16},  // 49
    {"branch-free K=1, ld=nt st=sc1, 14 wg/CU", 2 * kWaveBytes5, 64, 14},  // 50
    {"branch-free K=1, ld=nt st=sc1, 18 wg/CU", 2 * kWaveBytes5, 64, 18},  // 51
#endif
};
#ifdef CNT_LAB_VARIANTS
inline int encode2_waves(int variant) { return variant == 3 || (variant >= 38 && variant <= 40) ? 4 : variant == 2 || variant == 41 || variant == 42 ? 2 : 1; }
inline int encode2_pipe_k(int variant) { return variant >= 48 && variant <= 51 ? 1 : variant == 28 || variant == 29 || variant == 30 || variant == 36 ? 2 : variant == 34 || variant == 35 ? 8 : (variant >= 31 && variant <= 37) ? 4 : 0; }
#else
constexpr int encode2_waves(int) { return 1; }
constexpr int encode2_pipe_k(int) { return 0; }
constexpr int decode2_waves(int) { return 1; }
constexpr int decode2_pipe_k(int) { return 0; }
#endif
constexpr int kNumEncode2Variants = sizeof(kEncode2Variants) / sizeof(kEncode2Variants[0]);

constexpr VariantDesc kDecode2Variants[] = {
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-quads, ld=plain st=sc0|sc1|nt, 16 wg/CU", 2 * kWaveBytes5, 64, 16},  // 0: default
#ifdef CNT_LAB_VARIANTS
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt", 4 * kWaveBytes5, 64, 0},   // 1
    {"wave-tiled 2 words/lane, 2 waves/wg, ld=plain st=sc0|sc1|nt", 2 * kWaveBytes5, 64, 0},  // 2
    {"wave-tiled 2 words/lane, 4 waves/wg, ld=plain st=sc0|sc1|nt", 2 * kWaveBytes5, 64, 0},  // 3
    {"wave-tiled 1 word/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt (tiles split cache lines)", kWaveBytes5, 64, 0},  // 4
    {"wave-tiled 2 words/lane, 1 wave/wg, plain", 2 * kWaveBytes5, 64, 0},                    // 5
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt", 2 * kWaveBytes5, 64, 0},   // 6: as 0 without the residency cap
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt, 20 wg/CU", 2 * kWaveBytes5, 64, 20},  // 7
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt, 16 wg/CU", 2 * kWaveBytes5, 64, 16},  // 8
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt, 24 wg/CU", 2 * kWaveBytes5, 64, 24},  // 9: as 0 without the XCD pairing
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-pairs, ld=nt st=sc0|sc1|nt, 24 wg/CU", 2 * kWaveBytes5, 64, 24},  // 10
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-pairs, ld=plain st=sc0|sc1|nt, 15 wg/CU", 2 * kWaveBytes5, 64, 15},  // 11
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-pairs, ld=plain st=sc0|sc1|nt, 12 wg/CU", 2 * kWaveBytes5, 64, 12},  // 12
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt, 12 wg/CU", 2 * kWaveBytes5, 64, 12},  // 13
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt, 14 wg/CU", 2 * kWaveBytes5, 64, 14},  // 14
    {"wave-tiled 2 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt, 10 wg/CU", 2 * kWaveBytes5, 64, 10},  // 15
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-pairs, ld=plain st=sc0|sc1|nt, 18 wg/CU", 2 * kWaveBytes5, 64, 18},  // 16
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-pairs, ld=plain st=sc0|sc1|nt, 20 wg/CU", 2 * kWaveBytes5, 64, 20},  // 17
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-pairs, ld=plain st=sc0|sc1|nt, 24 wg/CU", 2 * kWaveBytes5, 64, 24},  // 18: the default before the caps were re-swept
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-pairs, ld=plain st=sc0|sc1|nt, 16 wg/CU", 2 * kWaveBytes5, 64, 16},  // 19: the default before the XCD group size was re-swept
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-quads, ld=plain st=sc0|sc1|nt, 15 wg/CU", 2 * kWaveBytes5, 64, 15},  // 20
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-quads, ld=plain st=sc0|sc1|nt, 18 wg/CU", 2 * kWaveBytes5, 64, 18},  // 21
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt, 7 wg/CU", 4 * kWaveBytes5, 64, 7},    // 22
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt, 8 wg/CU", 4 * kWaveBytes5, 64, 8},    // 23
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt, 9 wg/CU", 4 * kWaveBytes5, 64, 9},    // 24
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt, 10 wg/CU", 4 * kWaveBytes5, 64, 10},  // 25
    {"wave-tiled 4 words/lane, 1 wave/wg, ld=plain st=sc0|sc1|nt, 12 wg/CU", 4 * kWaveBytes5, 64, 12},  // 26
    {"wave-tiled 4 words/lane, 1 wave/wg, xcd-pairs, ld=plain st=sc0|sc1|nt, 8 wg/CU", 4 * kWaveBytes5, 64, 8},  // 27
    {"wave-tiled 4 words/lane, 1 wave/wg, xcd-pairs, ld=plain st=sc0|sc1|nt, 9 wg/CU", 4 * kWaveBytes5, 64, 9},  // 28
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-quads, ld=plain st=sc1|nt, 16 wg/CU", 2 * kWaveBytes5, 64, 16},   // 29
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-quads, ld=nt st=sc0|sc1|nt, 16 wg/CU", 2 * kWaveBytes5, 64, 16},  // 30
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-quads, ld=sc1 st=sc0|sc1|nt, 16 wg/CU", 2 * kWaveBytes5, 64, 16}, // 31
    // round 4: pipelined -- one wave takes K consecutive tiles and loads tile i+1's words before it expands tile i
    {"pipelined K=2, ld=plain st=sc0|sc1|nt, 16 wg/CU", 2 * kWaveBytes5, 64, 16},  // 32
    {"pipelined K=2, ld=plain st=sc0|sc1|nt, 12 wg/CU", 2 * kWaveBytes5, 64, 12},  // 33
    {"pipelined K=4, ld=plain st=sc0|sc1|nt, 16 wg/CU", 2 * kWaveBytes5, 64, 16},  // 34
    {"pipelined K=4, ld=plain st=sc0|sc1|nt, 12 wg/CU", 2 * kWaveBytes5, 64, 12},  // 35
    {"pipelined K=4, ld=plain st=sc0|sc1|nt, 10 wg/CU", 2 * kWaveBytes5, 64, 10},  // 36
    {"pipelined K=8, ld=plain st=sc0|sc1|nt, 12 wg/CU", 2 * kWaveBytes5, 64, 12},  // 37
    {"pipelined K=8, ld=plain st=sc0|sc1|nt, 8 wg/CU", 2 * kWaveBytes5, 64, 8},    // 38
    {"pipelined K=2, ld=plain st=sc0|sc1|nt, 20 wg/CU", 2 * kWaveBytes5, 64, 20},  // 39
    {"pipelined K=4, ld=plain st=sc0|sc1|nt, 20 wg/CU", 2 * kWaveBytes5, 64, 20},  // 40
    // four waves = one 4-KiB page of packed input per workgroup, under a residency cap for the first time
    {"wave-tiled 2 words/lane, 4 waves/wg, ld=plain st=sc0|sc1|nt, 3 wg/CU", 2 * kWaveBytes5, 64, 3},  // 41
    {"wave-tiled 2 words/lane, 4 waves/wg, ld=plain st=sc0|sc1|nt, 4 wg/CU", 2 * kWaveBytes5, 64, 4},  // 42
    {"wave-tiled 2 words/lane, 4 waves/wg, ld=plain st=sc0|sc1|nt, 5 wg/CU", 2 * kWaveBytes5, 64, 5},  // 43
    {"wave-tiled 2 words/lane, 2 waves/wg, ld=plain st=sc0|sc1|nt, 7 wg/CU", 2 * kWaveBytes5, 64, 7},  // 44
    {"wave-tiled 2 words/lane, 2 waves/wg, ld=plain st=sc0|sc1|nt, 8 wg/CU", 2 * kWaveBytes5, 64, 8},  // 45
    // 32 tiles per XCD turn: 27 whole pages of the WRITE stream (and 8 of the read stream) per turn
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-32s, ld=plain st=sc0|sc1|nt, 16 wg/CU", 2 * kWaveBytes5, 64, 16},  // 46
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-32s, ld=plain st=sc0|sc1|nt, 14 wg/CU", 2 * kWaveBytes5, 64, 14},  // 47
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-8s, ld=plain st=sc0|sc1|nt, 16 wg/CU", 2 * kWaveBytes5, 64, 16},   // 48
    {"wave-tiled 2 words/lane, 1 wave/wg, xcd-16s, ld=plain st=sc0|sc1|nt, 16 wg/CU", 2 * kWaveBytes5, 64, 16},  // 49
    // page tiles: one wave per 4-KiB page of the ASCII (write) stream, bits_to_n2_page
    {"page-tiled, plain order, ld=plain st=sc0|sc1|nt, 12 wg/CU", kPageNt5, 64, 12},  // 50
    {"page-tiled, plain order, ld=plain st=sc0|sc1|nt, 13 wg/CU", kPageNt5, 64, 13},  // 51
    {"page-tiled, plain order, ld=plain st=sc0|sc1|nt, 14 wg/CU", kPageNt5, 64, 14},  // 52
    {"page-tiled, plain order, ld=plain st=sc0|sc1|nt, 11 wg/CU", kPageNt5, 64, 11},  // 53
    {"page-tiled, plain order, ld=plain st=sc0|sc1|nt, 10 wg/CU", kPageNt5, 64, 10},  // 54
    {"page-tiled, plain order, ld=plain st=sc0|sc1|nt, 16 wg/CU", kPageNt5, 64, 16},  // 55
    {"page-tiled, xcd-quads, ld=plain st=sc0|sc1|nt, 12 wg/CU", kPageNt5, 64, 12},    // 56
    {"page-tiled, xcd-pairs, ld=plain st=sc0|sc1|nt, 12 wg/CU", kPageNt5, 64, 12},    // 57
    {"page-tiled, plain order, ld=nt st=sc0|sc1|nt, 12 wg/CU", kPageNt5, 64, 12},     // 58
    {"page-tiled, plain order, ld=plain st=sc0|sc1|nt, 18 wg/CU", kPageNt5, 64, 18},  // 59
    {"page-tiled, plain order, ld=plain st=sc0|sc1|nt, 20 wg/CU", kPageNt5, 64, 20},  // 60
    {"page-tiled, plain order, ld=plain st=sc0|sc1|nt, 24 wg/CU", kPageNt5, 64, 24},  // 61
    {"page-tiled, plain order, ld=plain st=sc0|sc1|nt, 30 wg/CU", kPageNt5, 64, 30},  // 62
    {"page-tiled, xcd-quads, ld=plain st=sc0|sc1|nt, 20 wg/CU", kPageNt5, 64, 20},    // 63
    // two pages per wave: 305 words in five rounds (95 % of the expansion lanes busy, against 80 % with one page)
    {"2-page-tiled, plain order, ld=plain st=sc0|sc1|nt, 6 wg/CU", 2 * kPageNt5, 64, 6},    // 64
    {"2-page-tiled, plain order, ld=plain st=sc0|sc1|nt, 7 wg/CU", 2 * kPageNt5, 64, 7},    // 65
    {"2-page-tiled, plain order, ld=plain st=sc0|sc1|nt, 8 wg/CU", 2 * kPageNt5, 64, 8},    // 66
    {"2-page-tiled, plain order, ld=plain st=sc0|sc1|nt, 9 wg/CU", 2 * kPageNt5, 64, 9},    // 67
    {"2-page-tiled, plain order, ld=plain st=sc0|sc1|nt, 10 wg/CU", 2 * kPageNt5, 64, 10},  // 68
    {"2-page-tiled, plain order, ld=plain st=sc0|sc1|nt, 12 wg/CU", 2 * kPageNt5, 64, 12},  // 69
    // branch-free twin of the word-tiled kernel (the pipelined kernel with K = 1; plain order)
    {"branch-free K=1, ld=plain st=sc0|sc1|nt, 16 wg/CU", 2 * kWaveBytes5, 64, 16},  // 70
    {"branch-free K=1, ld=plain st=sc0|sc1|nt, 14 wg/CU", 2 * kWaveBytes5, 64, 14},  // 71
    {"branch-free K=1, ld=plain st=sc0|sc1|nt, 18 wg/CU", 2 * kWaveBytes5, 64, 18},  // 72
#endif
};
constexpr int kFirstPageDecode2Variant = 50, kEndPageDecode2Variant = 70;
#ifdef CNT_LAB_VARIANTS
inline int decode2_waves(int variant) { return variant == 3 || (variant >= 41 && variant <= 43) ? 4 : variant == 2 || variant == 44 || variant == 45 ? 2 : 1; }
inline int decode2_pipe_k(int variant) { return variant >= 70 && variant <= 72 ? 1 : variant == 32 || variant == 33 || variant == 39 ? 2 : variant == 37 || variant == 38 ? 8 : (variant >= 34 && variant <= 40) ? 4 : 0; }
#endif
constexpr int kNumDecode2Variants = sizeof(kDecode2Variants) / sizeof(kDecode2Variants[0]);

// Whole wave tiles of [d_n, d_n + n_len) plus -- for the one-wave variants, in the same (last) launch -- the edge words `e`
// describes; *done_words = words the tiles cover (0: nothing was launched), *edges_done = whether the edges rode along
// (the two multi-wave variants, 2 and 3, leave them to the caller's generic launches).
template <bool STRICT>
int launch_encode2(int variant, const void* d_n, void* d_out, uint64_t n_len, Encode2Edges e, hipStream_t s, uint64_t* done_words, bool* edges_done,
                   BadCounter bad = BadCounter()) {
    if (bad) variant = 0;  // the checked twin exists for the shipped shape only
    if (variant < 0 || variant >= kNumEncode2Variants) return 1;
    const uint64_t tile_nt = kEncode2Variants[variant].tile_nt, tile_words = tile_nt / 27;
    const int pk = encode2_pipe_k(variant);
    const uint64_t total = pk ? n_len / tile_nt / pk * pk : n_len / tile_nt;  // pipelined: whole groups of K tiles, the rest are edge words
    *done_words = total * tile_words;
    const bool one_wave = encode2_waves(variant) == 1;
    *edges_done = one_wave && total > 0;
    e.tail_first = e.head_words + *done_words;
    const uint64_t per_launch = max_tiles_per_launch(64) / 8 * 8;  // wave tiles per launch (<= 2^31-1 threads; whole pipeline groups; the XCD maps are bijections for any count)
    const uint32_t xs = xcd_shift();
    for (uint64_t first = 0; first < total; first += per_launch) {
        const uint64_t n = total - first < per_launch ? total - first : per_launch;
        const uint8_t* in = static_cast<const uint8_t*>(d_n) + first * tile_nt;
        uint8_t* out = static_cast<uint8_t*>(d_out) + first * tile_words * 8;
        e.groups = (one_wave && first + n == total) ? edge_groups(e.head_words + (e.words - e.tail_first), 64, n) : 0u;
        // the static slab is 3488 B (2 words per lane) / 6944 B (4 words per lane), allocated in 512-B granules
        const uint32_t slab = encode2_waves(variant) == 4 ? 14336u : encode2_waves(variant) == 2 || tile_nt == 4 * kWaveBytes5 ? 7168u : 3584u;
        const uint32_t lds = lds_pad_for_cap(kEncode2Variants[variant].wg_cap, slab);
#define CNT_ENC2(W, P, L, S) \
    hipLaunchKernelGGL((n_to_bits2_wave<W, P, L, S, STRICT>), dim3(grid_of((n + W - 1) / W)), dim3(W * 64), lds, s, in, out, n, xs, e)
#define CNT_ENC2P(K, L, S) \
    hipLaunchKernelGGL((n_to_bits2_pipe<K, L, S, STRICT>), dim3(grid_of(n / K)), dim3(64), lds, s, in, out, e)
#ifdef CNT_LAB_VARIANTS
        if (pk) {
            e.groups = first + n == total ? edge_groups(e.head_words + (e.words - e.tail_first), 64, n / pk) : 0u;
            const uint32_t lds = lds_pad_for_cap(kEncode2Variants[variant].wg_cap, 4608u);  // the pipelined slab is a full 4 KiB (+16 B)
            if (pk == 1) CNT_ENC2P(1, kNT, kSC1);
            else if (variant == 37) CNT_ENC2P(4, kNT, kSC0 | kSC1 | kNT);
            else if (pk == 2) CNT_ENC2P(2, kNT, kSC1);
            else if (pk == 4) CNT_ENC2P(4, kNT, kSC1);
            else CNT_ENC2P(8, kNT, kSC1);
            continue;
        }
#endif
        if (bad) {
            hipLaunchKernelGGL((n_to_bits2_wave_checked<2, kNT, kSC1, STRICT>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e, bad.p, bad.mask);
            continue;
        }
        switch (variant) {
            case 0: CNT_ENC2(1, 2, kNT, kSC1); break;
#ifdef CNT_LAB_VARIANTS
            case 1: CNT_ENC2(1, 4, kNT, kSC1); break;
            case 2: case 41: case 42: CNT_ENC2(2, 2, kNT, kSC1); break;
            case 3: case 38: case 39: case 40: CNT_ENC2(4, 2, kNT, kSC1); break;
            case 4: CNT_ENC2(1, 1, kNT, kSC1); break;
            case 5: CNT_ENC2(1, 2, 0, 0); break;
            case 6: case 7: case 8: case 11: case 12: case 13: case 14: case 15: CNT_ENC2(1, 2, kNT, kSC1); break;
            case 16: case 17: case 18: case 19: case 20: CNT_ENC2(1, 4, kNT, kSC1); break;
            case 21: CNT_ENC2(1, 4, kNT, kSC0 | kSC1 | kNT); break;
            case 22: case 27: CNT_ENC2(1, 2, kNT, kSC0 | kSC1 | kNT); break;
            case 23: CNT_ENC2(1, 2, kNT, kSC1 | kNT); break;
            case 24: CNT_ENC2(1, 2, kNT, kNT); break;
            case 25: CNT_ENC2(1, 2, kSC1 | kNT, kSC1); break;
            case 26: CNT_ENC2(1, 2, 0, kSC1); break;
            case 9: hipLaunchKernelGGL((n_to_bits2_wave<1, 2, kNT, kSC1, STRICT, 2>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 10: hipLaunchKernelGGL((n_to_bits2_wave<1, 2, kNT, kSC0 | kSC1 | kNT, STRICT, 2>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 43: case 44: case 45: hipLaunchKernelGGL((n_to_bits2_wave<1, 2, kNT, kSC1, STRICT, 32>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 46: hipLaunchKernelGGL((n_to_bits2_wave<1, 2, kNT, kSC1, STRICT, 4>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 47: hipLaunchKernelGGL((n_to_bits2_wave<1, 2, kNT, kSC1, STRICT, 8>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
#endif
            default: return 1;
        }
#undef CNT_ENC2
#undef CNT_ENC2P
    }
    return 0;
}

// The any-alignment companion of encode2 variant 0: `base` = input pointer rounded down to 128 B,
// `phase` = the 1..127 bytes dropped.  Same LDS footprint per workgroup as variant 0 (15 wg/CU).
constexpr uint32_t kWindowEncode2Tile = 2 * kWaveBytes5;  // nt per tile (128 words)
constexpr uint32_t kWindowEncode2Slack = 128;             // bytes a tile may read behind its end
template <bool STRICT>
void launch_encode2_window(const uint8_t* base, uint32_t phase, uint8_t* out, uint64_t total_tiles, Encode2Edges e, hipStream_t s, BadCounter bad = BadCounter()) {
    const uint64_t per_launch = max_tiles_per_launch(64) / 4 * 4;
    const uint32_t xs = xcd_shift();
    const uint32_t lds = lds_pad_for_cap(kEncode2Variants[0].wg_cap, kWindowSlabDwords5 * 4u);  // four whole wave rows: 640 B more than variant 0's
    e.tail_first = e.head_words + total_tiles * (kWindowEncode2Tile / 27);
    for (uint64_t first = 0; first < total_tiles; first += per_launch) {
        const uint64_t n = total_tiles - first < per_launch ? total_tiles - first : per_launch;
        e.groups = first + n == total_tiles ? edge_groups(e.head_words + (e.words - e.tail_first), 64, n) : 0u;
        if (bad)
            hipLaunchKernelGGL((n_to_bits2_window_checked<kNT, kSC1, STRICT, 1>), dim3(grid_of(n)), dim3(64), lds, s,
                               base + first * kWindowEncode2Tile, out + first * 1024, n, phase, xs, e, bad.p, bad.mask);
        else
        hipLaunchKernelGGL((n_to_bits2_window<kNT, kSC1, STRICT, 1>), dim3(grid_of(n)), dim3(64), lds, s,
                           base + first * kWindowEncode2Tile, out + first * 1024, n, phase, xs, e);
    }
}

inline int launch_decode2(int variant, const void* d_bits, void* d_out, uint64_t len, Decode2Edges e, hipStream_t s, uint64_t* done_words, bool* edges_done) {
    if (variant < 0 || variant >= kNumDecode2Variants) return 1;
    const uint64_t tile_nt = kDecode2Variants[variant].tile_nt, tile_words = tile_nt / 27;
    const int pk = decode2_pipe_k(variant);
    const uint64_t total = pk ? len / tile_nt / pk * pk : len / tile_nt;
    *done_words = total * tile_words;
    const bool one_wave = decode2_waves(variant) == 1;
    *edges_done = one_wave && total > 0;
    e.tail_first = e.head_words + *done_words;
    const uint64_t per_launch = max_tiles_per_launch(64) / 8 * 8;
    const uint32_t xs = xcd_shift();
    constexpr int kAll = kSC0 | kSC1 | kNT;
    for (uint64_t first = 0; first < total; first += per_launch) {
        const uint64_t n = total - first < per_launch ? total - first : per_launch;
        const uint8_t* in = static_cast<const uint8_t*>(d_bits) + first * tile_words * 8;
        uint8_t* out = static_cast<uint8_t*>(d_out) + first * tile_nt;
        e.groups = (one_wave && first + n == total) ? edge_groups(e.head_words + (e.words - e.tail_first), 64, n) : 0u;
        const uint32_t slab = decode2_waves(variant) == 4 ? 14336u : decode2_waves(variant) == 2 || tile_nt == 4 * kWaveBytes5 ? 7168u : 3584u;  // static slab, see launch_encode2
        const uint32_t lds = lds_pad_for_cap(kDecode2Variants[variant].wg_cap, slab);
#define CNT_DEC2(W, P, L, S) \
    hipLaunchKernelGGL((bits_to_n2_wave<W, P, L, S>), dim3(grid_of((n + W - 1) / W)), dim3(W * 64), lds, s, in, out, n, xs, e)
#define CNT_DEC2P(K) \
    hipLaunchKernelGGL((bits_to_n2_pipe<K, 0, kAll>), dim3(grid_of(n / K)), dim3(64), lds, s, in, out, e)
#ifdef CNT_LAB_VARIANTS
        if (pk) {
            e.groups = first + n == total ? edge_groups(e.head_words + (e.words - e.tail_first), 64, n / pk) : 0u;
            const uint32_t lds = lds_pad_for_cap(kDecode2Variants[variant].wg_cap, 4608u);
            if (pk == 1) CNT_DEC2P(1);
            else if (pk == 2) CNT_DEC2P(2);
            else if (pk == 4) CNT_DEC2P(4);
            else CNT_DEC2P(8);
            continue;
        }
#endif
        switch (variant) {
            case 0: hipLaunchKernelGGL((bits_to_n2_wave<1, 2, 0, kAll, 4>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
#ifdef CNT_LAB_VARIANTS
            case 20: case 21: hipLaunchKernelGGL((bits_to_n2_wave<1, 2, 0, kAll, 4>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 19: case 11: case 12: case 16: case 17: case 18: hipLaunchKernelGGL((bits_to_n2_wave<1, 2, 0, kAll, 2>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 1: CNT_DEC2(1, 4, 0, kAll); break;
            case 2: case 44: case 45: CNT_DEC2(2, 2, 0, kAll); break;
            case 3: case 41: case 42: case 43: CNT_DEC2(4, 2, 0, kAll); break;
            case 4: CNT_DEC2(1, 1, 0, kAll); break;
            case 5: CNT_DEC2(1, 2, 0, 0); break;
            case 6: case 7: case 8: CNT_DEC2(1, 2, 0, kAll); break;
            case 9: case 13: case 14: case 15: CNT_DEC2(1, 2, 0, kAll); break;
            case 22: case 23: case 24: case 25: case 26: CNT_DEC2(1, 4, 0, kAll); break;
            case 27: case 28: hipLaunchKernelGGL((bits_to_n2_wave<1, 4, 0, kAll, 2>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 10: hipLaunchKernelGGL((bits_to_n2_wave<1, 2, kNT, kAll, 2>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 29: hipLaunchKernelGGL((bits_to_n2_wave<1, 2, 0, kSC1 | kNT, 4>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 30: hipLaunchKernelGGL((bits_to_n2_wave<1, 2, kNT, kAll, 4>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 31: hipLaunchKernelGGL((bits_to_n2_wave<1, 2, kSC1, kAll, 4>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 46: case 47: hipLaunchKernelGGL((bits_to_n2_wave<1, 2, 0, kAll, 32>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 48: hipLaunchKernelGGL((bits_to_n2_wave<1, 2, 0, kAll, 8>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
            case 49: hipLaunchKernelGGL((bits_to_n2_wave<1, 2, 0, kAll, 16>), dim3(grid_of(n)), dim3(64), lds, s, in, out, n, xs, e); break;
#endif
            default: return 1;
        }
#undef CNT_DEC2
#undef CNT_DEC2P
    }
    return 0;
}

#ifdef CNT_LAB_VARIANTS
// Page-tiled decode of a whole call: letters [0, head_nt) in front of the first 128-B line of d_out and the letters behind
// the last whole page ride as edge items in the (last) launch.  Returns 1 for an unknown variant, -1 when the call holds no
// whole page (the caller's generic kernel takes it), 0 after launching.
inline int launch_decode2_page(int variant, const uint64_t* bits, uint64_t words, uint8_t* out, uint64_t len, hipStream_t s) {
    if (variant < kFirstPageDecode2Variant || variant >= kEndPageDecode2Variant) return 1;
    const uint64_t tile_nt = kDecode2Variants[variant].tile_nt;
    const uint64_t head_nt = (128 - (reinterpret_cast<uintptr_t>(out) & 127)) & 127;
    if (len < head_nt + tile_nt) return -1;
    const uint64_t total = (len - head_nt) / tile_nt;
    Decode2PageEdges e{bits, out, len, head_nt, head_nt + total * tile_nt, 0};
    const uint64_t edge_items = (head_nt + 26) / 27 + (e.tail_from < len ? (len + 26) / 27 - e.tail_from / 27 : 0);
    const uint64_t per_launch = max_tiles_per_launch(64) / 8 * 8;
    const uint32_t xs = xcd_shift();
    const uint32_t lds = lds_pad_for_cap(kDecode2Variants[variant].wg_cap, (tile_nt == kPageNt5 ? PageTile5<1>::kSlabDwords : PageTile5<2>::kSlabDwords) * 4);
    constexpr int kAll = kSC0 | kSC1 | kNT;
    for (uint64_t first = 0; first < total; first += per_launch) {
        const uint64_t n = total - first < per_launch ? total - first : per_launch;
        e.groups = first + n == total ? edge_groups(edge_items, 64, n) : 0u;
        const uint64_t nt0 = head_nt + first * tile_nt;
#define CNT_DEC2PG(C, L, P) hipLaunchKernelGGL((bits_to_n2_page<C, L, kAll, P>), dim3(grid_of(n)), dim3(64), lds, s, bits, words, out, nt0, (uint32_t)n, xs, e)
        if (tile_nt != kPageNt5) { CNT_DEC2PG(1, 0, 2); continue; }
        switch (variant) {
            case 56: case 63: CNT_DEC2PG(4, 0, 1); break;
            case 57: CNT_DEC2PG(2, 0, 1); break;
            case 58: CNT_DEC2PG(1, kNT, 1); break;
            default: CNT_DEC2PG(1, 0, 1); break;
        }
#undef CNT_DEC2PG
    }
    return 0;
}
#endif

}  // namespace cnt
