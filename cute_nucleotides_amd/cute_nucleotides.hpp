// cute_nucleotides.hpp -- C++ host-side mirror of the reference crate's public API over the
// C ABI (include/cute_nt.h).  Header-only; link with -lcute_nt_hip.
//
// The reference exposes (src/lib.rs:1-2)
//     cute_nucleotides::n_to_bits::{n_to_bits_*, bits_to_n_*}      src/n_to_bits.rs
//     cute_nucleotides::n_to_bits2::{n_to_bits2_*, bits_to_n2_*}   src/n_to_bits2.rs
// as `fn(&[u8]) -> Vec<u64>` / `fn(&[u64], usize) -> Vec<u8>`.  The same names with the
// `_hip` suffix live here in the same module layout, take spans (pointer + length) and return
// owned vectors, exactly like the Rust functions: callee allocates, caller only lends.
//
// Error behaviour: `len` larger than the packed capacity throws std::length_error with the
// reference's panic text (n_to_bits.rs:52-54); any other failure (HIP error, no device)
// throws std::runtime_error carrying cnt_strerror().  Like the reference's functions these
// are re-entrant and callable from any thread.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/cute_nt.h"

namespace cute_nucleotides {

namespace detail {
inline void check(int status) {
    if (status == CNT_OK) return;
    if (status == CNT_ELEN) throw std::length_error(cnt_strerror(status));  // the reference's panic
    throw std::runtime_error(std::string("libcute_nt_hip: ") + cnt_strerror(status));
}
}  // namespace detail

namespace n_to_bits {

/// Encode {A,T/U,C,G} -> {00,10,01,11}, 32 nt per u64, LSB first (n_to_bits.rs:34-47 and the
/// four SIMD siblings).  `strict_lut` selects n_to_bits_lut's table semantics for bytes outside
/// the alphabet (they encode as 0) instead of the SIMD variants' (byte>>1)&3.
inline std::vector<uint64_t> n_to_bits_hip(const uint8_t* n, size_t len, bool strict_lut = false) {
    std::vector<uint64_t> out(cnt_words_for(len));
    detail::check(cnt_n_to_bits_ex(n, len, out.data(), out.size(), strict_lut ? CNT_STRICT_LUT : 0u));
    return out;
}
inline std::vector<uint64_t> n_to_bits_hip(const std::vector<uint8_t>& n) { return n_to_bits_hip(n.data(), n.size()); }

/// Decode `len` nucleotides (n_to_bits.rs:51-69 and the three SIMD siblings).
inline std::vector<uint8_t> bits_to_n_hip(const uint64_t* bits, size_t words, size_t len) {
    if (len > (words << 5)) detail::check(CNT_ELEN);
    std::vector<uint8_t> out(len);
    detail::check(cnt_bits_to_n(bits, words, len, out.data()));
    return out;
}
inline std::vector<uint8_t> bits_to_n_hip(const std::vector<uint64_t>& bits, size_t len) {
    return bits_to_n_hip(bits.data(), bits.size(), len);
}

/// The same, cut into contiguous chunks over `ndev` GPUs (<= 0: all visible), no collective.
inline std::vector<uint64_t> n_to_bits_hip_sharded(const uint8_t* n, size_t len, int ndev = 0) {
    std::vector<uint64_t> out(cnt_words_for(len));
    detail::check(cnt_n_to_bits_sharded(n, len, out.data(), out.size(), ndev));
    return out;
}
inline std::vector<uint8_t> bits_to_n_hip_sharded(const uint64_t* bits, size_t words, size_t len, int ndev = 0) {
    if (len > (words << 5)) detail::check(CNT_ELEN);
    std::vector<uint8_t> out(len);
    detail::check(cnt_bits_to_n_sharded(bits, words, len, out.data(), ndev));
    return out;
}

}  // namespace n_to_bits

namespace n_to_bits2 {

/// 5-letter codec, 3 nt -> 7 bits, 27 nt per u64 (n_to_bits2.rs:37-74, :118-189).
inline std::vector<uint64_t> n_to_bits2_hip(const uint8_t* n, size_t len) {
    std::vector<uint64_t> out(cnt_words2_for(len));
    detail::check(cnt_n_to_bits2(n, len, out.data(), out.size()));
    return out;
}
inline std::vector<uint64_t> n_to_bits2_hip(const std::vector<uint8_t>& n) { return n_to_bits2_hip(n.data(), n.size()); }

/// n_to_bits2.rs:78-107, :196-268.
inline std::vector<uint8_t> bits_to_n2_hip(const uint64_t* bits, size_t words, size_t len) {
    if (words > SIZE_MAX / 27 || len > words * 27) detail::check(CNT_ELEN);
    std::vector<uint8_t> out(len);
    detail::check(cnt_bits_to_n2(bits, words, len, out.data()));
    return out;
}
inline std::vector<uint8_t> bits_to_n2_hip(const std::vector<uint64_t>& bits, size_t len) {
    return bits_to_n2_hip(bits.data(), bits.size(), len);
}

}  // namespace n_to_bits2

}  // namespace cute_nucleotides
